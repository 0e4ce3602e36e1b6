#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats + PMC passes of the default bench.py
# command; leaves compact summaries under gpurun_out/profiles_<tag>/ for copying into profiles/.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cd "$R"
OUT=gpurun_out/profiles_$TAG
mkdir -p $OUT
BENCH="python bench.py --no-cpu-baseline"
# 1. un-profiled reference line
$BENCH > $OUT/bench_unprofiled.json 2> $OUT/bench_unprofiled.err
# 2. kernel trace + stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/bench_under_trace.json 2> $OUT/trace.err
cp $OUT/trace/*/*_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
# 3. PMC passes (each in its own run; no trace domains combined with --pmc)
for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -- python bench.py --steps 300 --warmup 150 --no-cpu-baseline > /dev/null 2> $OUT/pmc_$N.err
  python tools/pmc_summary.py $OUT/pmc_$N 150 > $OUT/pmc_$N.txt 2>&1
  rm -rf $OUT/pmc_$N
done
rm -rf $OUT/trace
tail -n +1 $OUT/kernel_stats.csv $OUT/pmc_*.txt | cut -c1-250
cat $OUT/bench_unprofiled.json
