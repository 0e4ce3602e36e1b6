#!/bin/bash
# Run on the GPU box (via gpurun): counters of the fused policy-trunk kernels at the c3 shapes (tools/mlp_kernels_bench.py),
# one --pmc group per pass: HBM traffic (FETCH_SIZE x2 on gfx950, WRITE_SIZE; KB units) and SQ instruction / activity counters.
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  rm -rf /tmp/pmc_o
  rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_o -- python tools/mlp_kernels_bench.py > /dev/null 2>&1
  python - <<'PY'
import csv,glob,collections
f=glob.glob('/tmp/pmc_o/*/*counter_collection.csv')[0]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
keys=('actor_l1_bwd_k','actor_l1_fwd_k','relu_ln_bwd_k','relu_ln_fwd_k','relu_ln_head_bwd_k','relu_ln_head_fwd_k')
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name']
    for key in keys:
        if key in k:
            name=key+('<HDP=0>' if ('l1_bwd' in key and ', 0>' in k) else '')
            acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
rows=4915200
for key,d in sorted(acc.items()):
    out=[]
    for c,v in d.items():
        m=sum(v)/len(v)
        if c in ('FETCH_SIZE','WRITE_SIZE'):
            gb=m*1024*(2 if c=='FETCH_SIZE' else 1)/1e9
            out.append("%s %.2f GB per launch%s"%(c,gb," (x2 gfx950 correction applied)" if c=='FETCH_SIZE' else ""))
        else:
            out.append("%s %.3g (%.1f per row)"%(c,m,m/rows))
    print("%-26s %s"%(key, "; ".join(out)))
PY
done
