#!/bin/bash
# Run on the GPU box (via gpurun): everything profiles/<round>/ is built from, ONE script for every round (it replaces the
# profile_round{,2,3,4,5}.sh / pmc_traffic_*.sh / harden_round5.sh family).  usage:
#     tools/profile_round.sh <round, e.g. r06> [stage ...]        (no stage: all of them, in this order)
# stages
#   bench     the driver's exact command (bench_driver_cmd.json) + rocprofv3 --kernel-trace --stats of the same c2 run
#             (kernel_stats_bench_default.csv, bench_under_trace.json: the roofline's launch time must agree with the trace)
#   shards    the per-GPU shards of c4 (16 x 256 x 1024) and c5 (32 x 1024 x 2048, pull force): bench_c4_shard.json, bench_c5_shard.json
#   c3        bench.py --mode mappo (shipped config): mappo_c3_default.json
#   traffic   HBM traffic by counter: rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE in SEPARATE passes of the rollout launches only
#             (DCC_AUTOTUNE=0, --place-tries 0) for c2 / c4 / c5, WRITE_SIZE calibrated on tools/membw.hip's store pattern,
#             FETCH_SIZE doubled (MI355X_MICROARCH.md, gfx950) -> traffic_c{2,4,5}.json (bench.py reads the newest round's files)
#   mfma      MFMA-pipe counters of the c3 iteration (tools/pmc_mfma_c3.sh) -> mappo_c3_mfma_pmc.txt
#   harden    host-side UBSan through the C-ABI GPU suites, graph-vs-eager rollout soak, both differential fuzzers -> harden/
#   train     tools/train_and_evaluate.py 5000 1024 -> training_run_5000_iters_1024_envs.txt (+ .json curve)   [~6 min]
# Output: gpurun_out/profiles_<round>/ (copy what is to be kept into profiles/<round>/).  --pmc passes are never combined with a
# trace domain (gpurun refuses that).
set -u
RND=${1:?usage: tools/profile_round.sh <round> [stage ...]}; shift
STAGES=${*:-bench shards c3 traffic mfma harden train}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cd "$R"
OUT=gpurun_out/profiles_$RND
mkdir -p $OUT
has() { [[ " $STAGES " == *" $1 "* ]]; }

if has bench; then
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --steps 20 --warmup 5 --no-c3 --no-cpu-baseline \
    > $OUT/bench_under_trace.json 2> $OUT/trace.err
  cp $OUT/trace/*/*_kernel_stats.csv $OUT/kernel_stats_bench_default.csv 2>/dev/null; rm -rf $OUT/trace
fi
if has shards; then
  python bench.py --agents 16 --pois 256 --envs 1024 --steps 5 --warmup 2 --launches-per-step 8 --no-c3 --no-cpu-baseline > $OUT/bench_c4_shard.json 2>/dev/null
  python bench.py --agents 32 --pois 1024 --envs 2048 --comm-force-scale 0.5 --r-comm 0.1 --steps-per-launch 50 --steps 3 --warmup 1 \
    --launches-per-step 2 --no-c3 --no-cpu-baseline > $OUT/bench_c5_shard.json 2>/dev/null
fi
if has c3; then
  python bench.py --mode mappo --iters 3 2>/dev/null | tail -1 > $OUT/mappo_c3_default.json
fi
if has traffic; then
  pmc_pass() {   # $1 = tag, $2 = steps per launch, rest = bench flags; only the timed / warm-up launches may reach the counters
    local tag=$1 T=$2; shift 2
    for C in WRITE_SIZE FETCH_SIZE; do
      rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${tag}_$C -- env DCC_AUTOTUNE=0 python bench.py --steps 1 --warmup 1 \
        --launches-per-step 4 --no-c3 --no-cpu-baseline --place-tries 0 "$@" > /dev/null 2> $OUT/pmc_${tag}_$C.err
      python tools/pmc_summary.py $OUT/pmc_${tag}_$C $T > $OUT/pmc_${tag}_$C.txt 2>&1
      rm -rf $OUT/pmc_${tag}_$C $OUT/pmc_${tag}_$C.err
    done
  }
  pmc_pass c2 150
  pmc_pass c4 30 --agents 16 --pois 256 --envs 8192 --steps-per-launch 30 --actions rng
  pmc_pass c5 4 --agents 32 --pois 1024 --envs 16384 --steps-per-launch 4 --actions rng --comm-force-scale 0.5 --r-comm 0.1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/membw.hip -o $OUT/membw 2> $OUT/membw_build.err && {
    $OUT/membw > $OUT/membw_fill_copy.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_cal -- $OUT/membw > /dev/null 2> $OUT/pmc_cal.err
    OUTDIR=$OUT python - <<'PY' > $OUT/pmc_WRITE_SIZE_calibration.txt
import csv, glob, collections, os
f = glob.glob(os.environ["OUTDIR"] + '/pmc_cal/**/*_counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r['Counter_Name'] == 'WRITE_SIZE':
        acc[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
known = 150 * 4096 * 676 * 16
for k, v in acc.items():
    per = sum(v) / len(v)
    if k.strip() in ('fill4', 'copy4', 'wave_blocks'):
        print("%-12s launches=%d  WRITE_SIZE=%.6g (KB units) -> %.4f GB per launch; known bytes written %.4f GB; calibration factor (known / counted) = %.4f"
              % (k.strip(), len(v), per, per * 1024 / 1e9, known / 1e9, known / (per * 1024)))
PY
    rm -rf $OUT/pmc_cal $OUT/membw $OUT/pmc_cal.err $OUT/membw_build.err
  }
  OUTDIR=$OUT RND=$RND python - <<'PY'
import json, os, re
out, rnd = os.environ["OUTDIR"] + "/", os.environ["RND"]
def per_launch(name):
    try:
        for l in open(out + name):
            m = re.search(r'total=([0-9.e+]+)', l)
            if m and 'n=' in l:
                return float(m.group(1))
    except Exception:
        pass
    return None
cal = 1.0
try:
    for l in open(out + 'pmc_WRITE_SIZE_calibration.txt'):
        if l.startswith('wave_blocks'):
            cal = float(l.rsplit('=', 1)[1])
except Exception:
    pass
src = ("rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE (separate passes of the rollout launches only: DCC_AUTOTUNE=0, --place-tries 0; KB units "
       "x1024; WRITE_SIZE calibrated on the known byte count of tools/membw.hip's wave_blocks store pattern, FETCH_SIZE doubled per "
       "MI355X_MICROARCH.md's gfx950 note), profiles/%s/pmc_%%s_WRITE_SIZE.txt, pmc_%%s_FETCH_SIZE.txt, pmc_WRITE_SIZE_calibration.txt "
       "(tools/profile_round.sh %s traffic)" % (rnd, rnd))
for tag, N, M, E, T, acts, alg in (("c2", 8, 64, 4096, 150, "hbm", 11851), ("c4", 16, 256, 8192, 30, "rng", 87563 - 128), ("c5", 32, 1024, 16384, 4, "rng", 676363 - 256)):
    w, f = per_launch('pmc_%s_WRITE_SIZE.txt' % tag), per_launch('pmc_%s_FETCH_SIZE.txt' % tag)
    if w is None or f is None:
        continue
    wb, fb = w * 1024 * cal, f * 1024 * 2
    json.dump({"workload": {"n_agents": N, "n_pois": M, "envs": E, "steps_per_launch": T, "actions": acts},
               "write_bytes_per_launch": wb, "write_size_calibration_factor": cal, "fetch_bytes_per_launch_corrected_x2": fb,
               "traffic_bytes_per_launch": wb + fb, "traffic_bytes_per_env_step": (wb + fb) / (E * T),
               "algorithmic_bytes_per_launch": alg * E * T, "algorithmic_bytes_per_env_step": alg, "source": src % (tag, tag)},
              open(out + 'traffic_%s.json' % tag, 'w'), indent=1)
PY
  cat $OUT/traffic_c2.json
fi
if has mfma; then
  bash tools/pmc_mfma_c3.sh > /dev/null 2>&1
  cp gpurun_out/pmc_mfma_c3.txt $OUT/mappo_c3_mfma_pmc.txt 2>/dev/null
  cp gpurun_out/pmc_mfma_c3/bench.json $OUT/mappo_c3_under_pmc.json 2>/dev/null
fi
if has harden; then
  mkdir -p $OUT/harden
  bash tools/ubsan_host_build.sh > $OUT/harden/ubsan_host.txt 2>&1; tail -3 gpurun_out/ubsan.log >> $OUT/harden/ubsan_host.txt
  python tools/graph_rollout_soak.py 150 4096 8 64 150 > $OUT/harden/graph_rollout_soak_c3.txt 2>&1
  python tools/graph_rollout_soak.py 400 64 4 20 50 > $OUT/harden/graph_rollout_soak_small.txt 2>&1
  python tools/fuzz_env_parity.py 240 556 > $OUT/harden/fuzz_env_parity.txt 2>&1
  python tools/fuzz_mlp_kernels.py 120 > $OUT/harden/fuzz_mlp_kernels.txt 2>&1
  tail -2 $OUT/harden/*.txt
fi
if has train; then
  python tools/train_and_evaluate.py 5000 1024 $OUT/training_curve_5000_iters_1024_envs.json > $OUT/training_run_5000_iters_1024_envs.txt 2>&1
  tail -12 $OUT/training_run_5000_iters_1024_envs.txt
fi
ls -la $OUT
