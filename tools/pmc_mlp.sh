#!/bin/bash
# Run on the GPU box (via gpurun): SQ instruction / activity counters of the fused policy-trunk kernels at the c3 shapes
# (tools/mlp_kernels_bench.py), two --pmc passes, printed per kernel.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_WR"; do
  rm -rf /tmp/pmc_o
  rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_o -- python tools/mlp_kernels_bench.py > /dev/null 2>&1
  python - <<'PY'
import csv,glob,collections
f=glob.glob('/tmp/pmc_o/*/*counter_collection.csv')[0]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name']
    for key in ('actor_l1_bwd','actor_l1_fwd','relu_ln_bwd_k','relu_ln_head_bwd'):
        if key in k:
            acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
for key,d in acc.items():
    print(key, {c: "%.3g"%(sum(v)/len(v)) for c,v in d.items()})
PY
done
