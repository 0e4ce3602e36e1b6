#!/bin/bash
# Run on the GPU box: kernel-trace stats of the c3 iteration with 5 and with 0 PPO epochs; the difference per kernel
# name is the update alone (3 iterations each: 2 warm-up + 1 timed).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cd "$R"
OUT=gpurun_out/prof_update_only
mkdir -p $OUT
for E in 5 0; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t$E -- python bench.py --mode mappo --iters 1 --ppo-epoch $E "$@" > $OUT/bench$E.json 2> $OUT/err$E.txt
  cp $OUT/t$E/*/*_kernel_stats.csv $OUT/kernel_stats_$E.csv; rm -rf $OUT/t$E
done
python - <<PY
import csv,re
def load(f):
    d={}
    for r in csv.DictReader(open(f)):
        d[r['Name']]=(int(r['TotalDurationNs'])/1e6,int(r['Calls']))
    return d
a=load('$OUT/kernel_stats_5.csv'); b=load('$OUT/kernel_stats_0.csv')
rows=[]
for k,(t,n) in a.items():
    t0,n0=b.get(k,(0,0))
    rows.append((t-t0,n-n0,k))
rows.sort(reverse=True)
tot=sum(r[0] for r in rows)
print("update-only kernel time: %.1f ms for 15 epochs = %.2f ms/epoch; rollout-only %.1f ms per 3 rollouts"%(tot,tot/15,sum(v[0] for v in b.values())))
for t,n,k in rows[:45]:
    short=re.sub(r'at::native::|\(anonymous namespace\)::|void ','',k)[:95]
    print("%7.2f ms/epoch  %5.1f calls/epoch  %s"%(t/15,n/15,short))
small=sum(t for t,n,k in rows if n>0 and t/n<0.05)
print("kernels under 50 us each: %.2f ms/epoch in %.0f calls/epoch"%(small/15, sum(n for t,n,k in rows if n>0 and t/n<0.05)/15))
PY
