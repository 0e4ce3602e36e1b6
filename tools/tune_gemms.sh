#!/bin/bash
# Regenerates dynamic-coverage-control_amd/config/gemm_tunings_gfx950.csv: one MAPPO iteration per BASELINE shape with torch's
# TunableOp tuning every library GEMM it meets (eager rollout: nothing may be tuned under stream capture), results merged
# into one file.  Run on the GPU box (about 15 minutes): bash tools/tune_gemms.sh
set -e
cd "$(dirname "$0")/.."
OUT=gpurun_out/gemm_tunings
mkdir -p gpurun_out
rm -f ${OUT}*.csv
export PYTORCH_TUNABLEOP_FILENAME=$PWD/${OUT}.csv PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=${TUNE_MS:-30} PYTORCH_TUNABLEOP_MAX_TUNING_ITERATIONS=${TUNE_ITERS:-5}
run() { timeout ${TUNE_TIMEOUT:-1500} python bench.py --mode mappo --iters 1 --no-graph "$@" > gpurun_out/tune_gemms.log 2>&1 || tail -5 gpurun_out/tune_gemms.log; wc -l ${OUT}0.csv; }
run                                                        # c3: 8 x 64 x 4096 envs
run --agents 16 --pois 256 --envs 1024                     # c4 per-GPU shard
run --agents 32 --pois 1024 --envs 256 --comm-force-scale 0.5 --r-comm 0.1   # c5 shape, 256 envs
run --agents 32 --pois 1024 --envs 2048 --comm-force-scale 0.5 --r-comm 0.1  # c5 per-GPU shard (two chunks per epoch)
cp ${OUT}0.csv ${TUNE_DEST:-dynamic-coverage-control_amd/config/gemm_tunings_gfx950.csv}
