#!/bin/bash
# Which hardware counter tells a slow placement of the output buffer from a fast one?  tools/placement_probe3.py streams the
# observation producer (K = 150) into 10 separately allocated buffers, 3 launches each; rocprofv3 --pmc per counter group; per
# buffer: mean duration and mean counter values.  Run on the GPU box.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cd "$R"
OUT=gpurun_out/placement_pmc
rm -rf $OUT; mkdir -p $OUT
for G in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" \
         "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
         "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCC_EA0_WRREQ_LEVEL_sum" \
         "TCC_TAG_STALL_sum TCC_IB_STALL_sum TCC_BUSY_sum" \
         "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum"; do
  N=$(echo $G | tr ' ' '+' | cut -c1-60)
  rocprofv3 --pmc $G --output-format csv -d $OUT/p -- python tools/placement_probe3.py > /dev/null 2> $OUT/err.txt
  python - "$OUT/p" "$G" <<'PY'
import csv, glob, sys, collections
d, names = sys.argv[1], sys.argv[2].split()
f = glob.glob(d + '/**/*_counter_collection.csv', recursive=True)[0]
rows = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if 'dcc_env_roles_kernel' not in r['Kernel_Name']:
        continue
    k = int(r['Dispatch_Id'])
    e = rows.setdefault(k, {'dur': (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6})
    e[r['Counter_Name']] = float(r['Counter_Value'])
disp = [v for k, v in sorted(rows.items()) if v['dur'] > 0.9]          # the K = 150 launches only
per = [disp[i:i + 3] for i in range(0, len(disp) - len(disp) % 3, 3)]
print("counter group:", " ".join(names))
for i, g in enumerate(per):
    print("  buffer %2d  dur %.4f ms  " % (i, sum(x['dur'] for x in g) / 3) + "  ".join("%s=%.4g" % (n.replace('_sum', ''), sum(x.get(n, x.get(n.replace('_sum',''), 0)) for x in g) / 3) for n in names))
PY
  rm -rf $OUT/p
done
