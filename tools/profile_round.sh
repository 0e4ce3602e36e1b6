#!/bin/bash
# Run on the GPU box (via gpurun): everything profiles/<tag>/ is built from, in one call.
#   1. tools/profile_bench.sh <tag>      headline env-step bench: un-profiled line, kernel-trace stats, PMC passes
#   2. bench.py --mode mappo             c3 end-to-end lines: default config, structured input, compact rows
#   3. tools/profile_mappo.sh            kernel-trace stats of the structured c3 iteration
#   4. tools/mlp_kernels_bench.py        the fused policy-trunk kernels at the c3 shapes
#   5. a 60-iteration training run of the shipped task on the fast path
# Output: gpurun_out/profiles_<tag>/ (copy what is to be kept into profiles/<tag>/).
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
OUT=gpurun_out/profiles_$TAG
mkdir -p $OUT
tools/profile_bench.sh $TAG > $OUT/profile_bench.log 2>&1
python bench.py > $OUT/bench_default_with_cpu_baseline.json 2> $OUT/bench_default.err
python bench.py --steps-per-launch 1 --steps 3000 --warmup 300 --no-cpu-baseline > $OUT/bench_single_step_launches.json 2>/dev/null
DCC_NO_ROLES=1 python bench.py --no-cpu-baseline > $OUT/bench_fused_kernel.json 2>/dev/null
for V in "default:" "state_only:--compact-obs" "dense_rows:--no-structured-input" "dense_rows_eager:--no-structured-input --no-graph" "dense_compact:--no-structured-input --compact-obs --update-chunk-steps 10"; do
  NAME=${V%%:*}; FLAGS=${V#*:}
  python bench.py --mode mappo --iters 3 $FLAGS 2>/dev/null | tail -1 > $OUT/mappo_c3_$NAME.json
done
tools/profile_mappo.sh ${TAG}_mappo_default > $OUT/mappo_c3_default_profile.txt 2>&1
cp gpurun_out/prof_${TAG}_mappo_default/kernel_stats.csv $OUT/mappo_c3_default_kernel_stats.csv
tools/profile_mappo.sh ${TAG}_mappo_dense --no-structured-input > $OUT/mappo_c3_dense_rows_profile.txt 2>&1
cp gpurun_out/prof_${TAG}_mappo_dense/kernel_stats.csv $OUT/mappo_c3_dense_rows_kernel_stats.csv
tools/profile_update_only.sh > $OUT/mappo_c3_update_only.txt 2>&1
python tools/mlp_kernels_bench.py > $OUT/mlp_kernels.txt 2>/dev/null
( cd dynamic-coverage-control_amd && python train.py 0 n_iters=60 n_rollout_threads=1024 n_eval_rollout_threads=0 save_model=False log_interval=5 ) > $OUT/training_run_structured.log 2>&1
ls -la $OUT
for f in $OUT/mappo_c3_*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read()); c=d['config']
print(round(d['value']), 'rollout %.4f update %.4f peak %.1f GB'%(c['rollout_s_per_iter'], c['update_s_per_iter'], c['peak_hbm_gb']))"; done
cat $OUT/bench_default_with_cpu_baseline.json
cat $OUT/mlp_kernels.txt
tail -15 $OUT/training_run_structured.log
