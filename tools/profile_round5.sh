#!/bin/bash
# Run on the GPU box (via gpurun): what profiles/r05/ is built from, in one call on ONE box.  The c2 / c4 kernels did not change this
# round (VERDICT r04 stop list), so the counter passes of round 4 (profiles/r04/traffic_c{2,4,5}.json) stay the traffic source.
#   1. the driver's exact bench command; rocprofv3 --kernel-trace --stats of the same c2 run (the roofline's launch time must agree)
#   2. the per-GPU shards: c4 (16 x 256 x 1024), c5 (32 x 1024 x 2048, pull force) -- c5 three times each as shipped and with
#      DCC_NO_SPEC=1 (generic runtime-size kernel), interleaved on this box: the round-5 (32,1024) specialisation A/B
#   3. c3: bench.py --mode mappo (shipped config), and the same with num_mini_batch 2 / 4 (--num-mini-batch)
# Output: gpurun_out/profiles_r05/ (copy what is to be kept into profiles/r05/).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cd "$R"
OUT=gpurun_out/profiles_r05
rm -rf $OUT; mkdir -p $OUT
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err
C2="python bench.py --steps 20 --warmup 5 --no-c3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $C2 > $OUT/bench_under_trace.json 2> $OUT/trace.err
cp $OUT/trace/*/*_kernel_stats.csv $OUT/kernel_stats_bench_default.csv 2>/dev/null; rm -rf $OUT/trace
C5="python bench.py --agents 32 --pois 1024 --envs 2048 --comm-force-scale 0.5 --r-comm 0.1 --steps-per-launch 50 --steps 3 --warmup 1 --launches-per-step 2 --no-c3 --no-cpu-baseline"
for i in 1 2 3; do
  $C5 > $OUT/bench_c5_shard_spec_$i.json 2>/dev/null
  DCC_NO_SPEC=1 $C5 > $OUT/bench_c5_shard_generic_$i.json 2>/dev/null
done
C5L="python bench.py --agents 32 --pois 1024 --envs 16384 --comm-force-scale 0.5 --r-comm 0.1 --steps-per-launch 4 --steps 6 --warmup 1 --launches-per-step 4 --no-c3 --no-cpu-baseline --place-tries 1"
for i in 1 2; do
  $C5L > $OUT/bench_c5_leg_spec_$i.json 2>/dev/null
  DCC_NO_SPEC=1 $C5L > $OUT/bench_c5_leg_generic_$i.json 2>/dev/null
done
python - <<'PY' > $OUT/c5_spec_ab.txt
import json, glob
for tag in ("shard_spec", "shard_generic", "leg_spec", "leg_generic"):
    v = []
    for f in sorted(glob.glob("gpurun_out/profiles_r05/bench_c5_%s_*.json" % tag)):
        try:
            d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]
            v.append((r["frac"], r["launch_ms_avg"], r["first_allocation"]["frac"] if r.get("first_allocation") else None))
        except Exception as e:
            v.append(("error", str(e)[:60], None))
    print(tag, v)
PY
python bench.py --agents 16 --pois 256 --envs 1024 --steps 5 --warmup 2 --launches-per-step 8 --no-c3 --no-cpu-baseline > $OUT/bench_c4_shard.json 2>/dev/null
python bench.py --mode mappo --iters 3 2>/dev/null | tail -1 > $OUT/mappo_c3_default.json
python bench.py --mode mappo --iters 2 --num-mini-batch 2 2>/dev/null | tail -1 > $OUT/mappo_c3_mini_batch_2.json
python bench.py --mode mappo --iters 2 --num-mini-batch 4 2>/dev/null | tail -1 > $OUT/mappo_c3_mini_batch_4.json
ls -la $OUT
cat $OUT/c5_spec_ab.txt
