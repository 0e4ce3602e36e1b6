#!/bin/bash
# Run on the GPU box (via gpurun): SQ instruction / activity counters of the c2 headline launch (rollout launches only), one --pmc
# group per pass -> gpurun_out/pmc_env_sq.txt: instructions per env-step by class, and how busy the VALU / LDS / store issue is.
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/pmc_env_sq.txt; : > $OUT
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAVES" "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pmc_o
  rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_o -- env DCC_AUTOTUNE=0 python bench.py --steps 1 --warmup 1 --launches-per-step 4 --no-c3 --no-cpu-baseline --place-tries 0 $@ > /dev/null 2> /tmp/pmc_err.txt || tail -3 /tmp/pmc_err.txt >> $OUT
  python - >> $OUT <<'PY'
import csv,glob,collections
fs=glob.glob('/tmp/pmc_o/**/*counter_collection.csv', recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in fs:
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','')
        if 'dcc_env' in k:
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
steps=4096*150
for k,d in sorted(acc.items()):
    for c,v in d.items():
        m=sum(v)/len(v)
        print("%-60s %-26s n=%d mean=%.6g  per env-step %.2f" % (k[:44], c, len(v), m, m/steps))
PY
done
cat $OUT
