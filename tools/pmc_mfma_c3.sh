#!/bin/bash
# Run on the GPU box (via gpurun): MFMA-pipe counters of the SHIPPED c3 iteration (bench.py --mode mappo: policy forward in
# the rollout + PPO update), one rocprofv3 --pmc pass (no trace domains combined with it), aggregated per kernel name.
#   SQ_VALU_MFMA_BUSY_CYCLES  cycles an MFMA pipe was busy, summed over the SIMDs of all XCDs
#   SQ_BUSY_CYCLES            SQ busy cycles (summed over the SEs / XCDs that report)
#   SQ_INSTS_VALU_MFMA_MOPS_F32   f32 MFMA work in units of 512 FLOP
#   GRBM_GUI_ACTIVE           GPU-active cycles, reported as the SUM over the 8 XCDs
# Utilisation of the MFMA pipes = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024 SIMDs); cross-check = MOPS * 512 FLOP / kernel time
# against the 157.3 TFLOP/s fp32 matrix peak (kernel times: the kernel-trace run next to it, tools/profile_update_only.sh).
# usage: tools/pmc_mfma_c3.sh [bench flags...]  -> gpurun_out/pmc_mfma_c3.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
OUT=gpurun_out/pmc_mfma_c3
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/p -- \
  python bench.py --mode mappo --iters 1 --ppo-epoch 2 "$@" > $OUT/bench.json 2> $OUT/err.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- \
  python bench.py --mode mappo --iters 1 --ppo-epoch 2 "$@" > $OUT/bench_trace.json 2>> $OUT/err.txt
python - <<'PY' | tee gpurun_out/pmc_mfma_c3.txt
import csv, glob, collections, re
f = glob.glob('gpurun_out/pmc_mfma_c3/p/*/*counter_collection.csv')[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    key = (r.get('Dispatch_Id') or r.get('Correlation_Id'), k)
    if key not in seen:
        seen.add(key); calls[k] += 1
dur = {}
for st in glob.glob('gpurun_out/pmc_mfma_c3/t/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(st)):
        dur[r['Name']] = (float(r['TotalDurationNs']), int(r['Calls']))
print("c3 iteration under --pmc (3 iterations: 150-step rollout + 2 PPO epochs each); fp32 matrix peak 157.3 TFLOP/s")
print("%-74s %6s %9s %9s %8s %9s" % ("kernel", "calls", "mfma_util", "sq_busy", "TF/s", "ms/call"))
rows = []
for k, d in acc.items():
    gui = d.get('GRBM_GUI_ACTIVE', 0.0)
    busy = d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
    mops = d.get('SQ_INSTS_VALU_MFMA_MOPS_F32', 0.0)
    util = busy / (gui / 8.0 * 1024.0) if gui else 0.0
    sq = d.get('SQ_BUSY_CYCLES', 0.0) / gui if gui else 0.0
    t, n = dur.get(k, (0.0, 0))
    tf = (mops * 512.0 / calls[k]) / (t / n * 1e-9) / 1e12 if (n and t) else 0.0
    rows.append((gui, k, calls[k], util, sq, tf, t / n / 1e6 if n else 0.0))
rows.sort(reverse=True)
tot_gui = sum(r[0] for r in rows)
for gui, k, n, util, sq, tf, ms in rows[:28]:
    short = re.sub(r'at::native::|\(anonymous namespace\)::|void ', '', k)[:72]
    print("%-74s %6d %9.3f %9.2f %8.1f %9.4f   (%.1f%% of GPU-active cycles)" % (short, n, util, sq, tf, ms, 100 * gui / tot_gui))
gemm = [r for r in rows if r[1].startswith('Cijk')]
g_busy = sum(acc[r[1]]['SQ_VALU_MFMA_BUSY_CYCLES'] for r in gemm); g_gui = sum(r[0] for r in gemm)
all_busy = sum(d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) for d in acc.values())
print("all library GEMMs (Cijk_*): MFMA pipes busy %.3f of their GPU-active time; whole iteration: %.3f (GEMMs are %.1f%% of GPU-active cycles)"
      % (g_busy / (g_gui / 8 * 1024), all_busy / (tot_gui / 8 * 1024), 100 * g_gui / tot_gui))
PY
