#!/bin/bash
# Run on the GPU box: kernel-trace stats of c3 rollouts only (0 PPO epochs; 3 rollouts: eager, capture pass, one replay):
# per-kernel average of what one rollout step launches.  Output: gpurun_out/prof_rollout_only/summary.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cd "$R"
OUT=gpurun_out/prof_rollout_only
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python bench.py --mode mappo --iters 1 --ppo-epoch 0 "$@" > $OUT/bench.json 2> $OUT/err.txt
cp $OUT/t/*/*_kernel_stats.csv $OUT/kernel_stats.csv; rm -rf $OUT/t
python - <<PY > $OUT/summary.txt
import csv, re
rows = []
for r in csv.DictReader(open('$OUT/kernel_stats.csv')):
    rows.append((int(r['TotalDurationNs']) / 1e3, int(r['Calls']), r['Name']))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("rollout-only kernel time %.1f ms for 3 rollouts of 150 steps = %.1f us per env step" % (tot / 1e3, tot / 450))
for t, n, k in rows[:28]:
    short = re.sub(r'at::native::|\(anonymous namespace\)::|void ', '', k)[:110]
    print("%8.1f us/step %6.2f calls/step %8.1f us avg  %s" % (t / 450, n / 450, t / n, short))
PY
cat $OUT/summary.txt
