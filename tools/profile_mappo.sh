#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats of the c3 MAPPO iteration (bench.py --mode mappo).
# usage: tools/profile_mappo.sh <tag> [bench flags...]   -> gpurun_out/prof_<tag>/{kernel_stats.csv,bench.json}
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cd "$R"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python bench.py --mode mappo --iters 1 --ppo-epoch 5 "$@" > $OUT/bench.json 2> $OUT/err.txt
cp $OUT/t/*/*_kernel_stats.csv $OUT/kernel_stats.csv
rm -rf $OUT/t
python - <<PY
import csv,re
rows=list(csv.DictReader(open('$OUT/kernel_stats.csv')))
tot=sum(int(r['TotalDurationNs']) for r in rows)
print("total kernel ms %.1f (3 iterations x (150-step rollout + 5 epochs))"%(tot/1e6))
for r in rows[:26]:
    short=re.sub(r'at::native::|\(anonymous namespace\)::|void ','',r['Name'])[:100]
    print("%7.1f ms %5s%% calls %5s avg %8.1f us  %s"%(int(r['TotalDurationNs'])/1e6, r['Percentage'][:5], r['Calls'], float(r['AverageNs'])/1e3, short))
PY
tail -c 700 $OUT/bench.json
