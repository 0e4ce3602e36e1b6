#!/bin/bash
# Run on the GPU box (via gpurun): everything profiles/r02/ is built from, in one call on one box.
#   1. the driver's exact bench command (c2 headline + cpu_baseline + bounded c3 leg)
#   2. rocprofv3 --kernel-trace --stats of the same c2 run (env-kernel average must agree with the HIP-event figure)
#   3. HBM traffic of the env kernel: --pmc WRITE_SIZE and --pmc FETCH_SIZE in SEPARATE passes (no trace domains), plus a
#      CALIBRATION of WRITE_SIZE on a known byte count in the same store pattern (tools/membw.hip: 6.64 GB per launch), as
#      MI355X_MICROARCH.md's HBM section prescribes; FETCH_SIZE doubled (gfx950 correction for wide streaming reads)
#   4. c3: bench.py --mode mappo lines, kernel-trace stats of the iteration, update-only split, MFMA PMC pass
#   5. the fused policy-trunk kernels at the c3 shapes; a training run of the shipped task + evaluation
# Output: gpurun_out/profiles_r02/ (copy what is to be kept into profiles/r02/).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cd "$R"
OUT=gpurun_out/profiles_r02
rm -rf $OUT; mkdir -p $OUT
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err
C2="python bench.py --steps 20 --warmup 5 --no-c3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $C2 > $OUT/bench_under_trace.json 2> $OUT/trace.err
cp $OUT/trace/*/*_kernel_stats.csv $OUT/kernel_stats_bench_default.csv 2>/dev/null; rm -rf $OUT/trace
DCC_NO_ROLES=1 $C2 > $OUT/bench_fused_kernel.json 2>/dev/null
# --- PMC passes of the env kernel: 8 launches of 150 steps
PM="python bench.py --steps 1 --warmup 1 --launches-per-step 4 --no-c3 --no-cpu-baseline"
for C in WRITE_SIZE FETCH_SIZE "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -- $PM > /dev/null 2> $OUT/pmc_$N.err
  python tools/pmc_summary.py $OUT/pmc_$N 150 > $OUT/pmc_$N.txt 2>&1
  rm -rf $OUT/pmc_$N $OUT/pmc_$N.err
done
# --- WRITE_SIZE calibration on a known byte count (fill / the env kernel's store pattern: 150*4096*676*16 B per launch)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/membw.hip -o $OUT/membw 2> $OUT/membw_build.err && {
  $OUT/membw > $OUT/membw_fill_copy.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_cal -- $OUT/membw > /dev/null 2> $OUT/pmc_cal.err
  python - <<'PY' > $OUT/pmc_WRITE_SIZE_calibration.txt
import csv, glob, collections
f = glob.glob('gpurun_out/profiles_r02/pmc_cal/**/*_counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r['Counter_Name'] == 'WRITE_SIZE':
        acc[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
known = 150 * 4096 * 676 * 16
for k, v in acc.items():
    per = sum(v) / len(v)
    per_byte = {'fill4': known, 'copy4': known, 'wave_blocks': known}.get(k.strip(), None)
    if per_byte:
        print("%-12s launches=%d  WRITE_SIZE=%.6g (KB units) -> %.4f GB per launch; known bytes written %.4f GB; calibration factor (known / counted) = %.4f"
              % (k.strip(), len(v), per, per * 1024 / 1e9, per_byte / 1e9, per_byte / (per * 1024)))
PY
  rm -rf $OUT/pmc_cal $OUT/membw $OUT/pmc_cal.err
}
python - <<'PY'
import json, re
out = 'gpurun_out/profiles_r02/'
def per_launch(name):
    for l in open(out + name):         # tools/pmc_summary.py line of the 150-step launches: "... <COUNTER>  n=8  total=6.546e+06 ..."
        m = re.search(r'total=([0-9.e+]+)', l)
        if m and 'n=' in l:
            return float(m.group(1))
    return None
w, f = per_launch('pmc_WRITE_SIZE.txt'), per_launch('pmc_FETCH_SIZE.txt')
cal = 1.0
try:
    for l in open(out + 'pmc_WRITE_SIZE_calibration.txt'):
        if l.startswith('wave_blocks'):
            cal = float(l.rsplit('=', 1)[1])
except Exception:
    pass
if w is not None and f is not None:
    wb, fb = w * 1024 * cal, f * 1024 * 2
    json.dump({"workload": {"n_agents": 8, "n_pois": 64, "envs": 4096, "steps_per_launch": 150, "actions": "hbm"},
               "write_bytes_per_launch": wb, "write_size_calibration_factor": cal, "fetch_bytes_per_launch_corrected_x2": fb,
               "traffic_bytes_per_launch": wb + fb, "algorithmic_bytes_per_launch": 11851 * 4096 * 150,
               "source": "rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE (separate passes, KB units x1024; WRITE_SIZE calibrated on the "
                         "known byte count of tools/membw.hip's wave_blocks store pattern, FETCH_SIZE doubled per MI355X_MICROARCH.md's "
                         "gfx950 note), profiles/r02/pmc_WRITE_SIZE.txt, pmc_FETCH_SIZE.txt, pmc_WRITE_SIZE_calibration.txt"},
              open(out + 'traffic_c2.json', 'w'), indent=1)
PY
# --- c3
for V in "default:" "rows_stored:--keep-rows" "dense_rows:--no-structured-input" "dense_rows_eager:--no-structured-input --no-graph"; do
  NAME=${V%%:*}; FLAGS=${V#*:}
  python bench.py --mode mappo --iters 3 $FLAGS 2>/dev/null | tail -1 > $OUT/mappo_c3_$NAME.json
done
DCC_TUNED_GEMMS=0 python bench.py --mode mappo --iters 3 2>/dev/null | tail -1 > $OUT/mappo_c3_untuned_gemms.json
python tools/gae_time.py > $OUT/gae_kernel.txt 2>/dev/null
tools/profile_mappo.sh r02_mappo_default > $OUT/mappo_c3_default_top_kernels.txt 2>&1
cp gpurun_out/prof_r02_mappo_default/kernel_stats.csv $OUT/mappo_c3_default_kernel_stats.csv
tools/profile_update_only.sh > $OUT/mappo_c3_update_only.txt 2>&1
cp gpurun_out/prof_update_only/kernel_stats_0.csv $OUT/mappo_c3_rollout_only_kernel_stats.csv
tools/pmc_mfma_c3.sh > /dev/null 2>&1; cp gpurun_out/pmc_mfma_c3.txt $OUT/mappo_c3_mfma_pmc.txt
python tools/mlp_kernels_bench.py > $OUT/mlp_kernels.txt 2>/dev/null
# --- other BASELINE shapes through the same bench (per-GPU shards)
python bench.py --agents 16 --pois 256 --envs 1024 --steps 5 --warmup 2 --launches-per-step 8 --no-c3 --no-cpu-baseline > $OUT/bench_c4_shard.json 2>/dev/null
python bench.py --agents 32 --pois 1024 --envs 2048 --comm-force-scale 0.5 --r-comm 0.1 --steps-per-launch 50 --steps 3 --warmup 1 --launches-per-step 2 --no-c3 --no-cpu-baseline > $OUT/bench_c5_shard.json 2>/dev/null
python bench.py --agents 4 --pois 16 --steps 5 --warmup 2 --no-c3 --no-cpu-baseline > $OUT/bench_c1_size.json 2>/dev/null
# --- learning run of the shipped task
python tools/train_and_evaluate.py 300 1024 > $OUT/training_run_shipped_evaluate.txt 2>&1
ls -la $OUT
cat $OUT/pmc_WRITE_SIZE_calibration.txt $OUT/traffic_c2.json
tail -4 $OUT/training_run_shipped_evaluate.txt
