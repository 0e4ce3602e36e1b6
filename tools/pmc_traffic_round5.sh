#!/bin/bash
# Run on the GPU box: HBM traffic by counter for the FINAL round-5 binary (one env-kernel change this round: the reward's wave sum).
# rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE in separate passes (+ the WRITE_SIZE calibration on tools/membw.hip's store pattern) for c2
# (per launch) and the c4 / c5 shapes -> traffic_c2.json, traffic_c4.json, traffic_c5.json (bench.py reads the newest round's files).
# Output: gpurun_out/profiles_r05_pmc/ (copy into profiles/r05/).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cd "$R"
OUT=gpurun_out/profiles_r05_pmc
rm -rf $OUT; mkdir -p $OUT
# --- counters: only the timed / warm-up launches may reach them (no create-time shape measurement, no placement probes)
pmc_pass() {   # $1 = tag, $2 = steps per launch, rest = bench flags
  local tag=$1 T=$2; shift 2
  for C in WRITE_SIZE FETCH_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${tag}_$C -- env DCC_AUTOTUNE=0 python bench.py --steps 1 --warmup 1 --launches-per-step 4 --no-c3 --no-cpu-baseline --place-tries 0 "$@" > /dev/null 2> $OUT/pmc_${tag}_$C.err
    python tools/pmc_summary.py $OUT/pmc_${tag}_$C $T > $OUT/pmc_${tag}_$C.txt 2>&1
    rm -rf $OUT/pmc_${tag}_$C $OUT/pmc_${tag}_$C.err
  done
}
pmc_pass c2 150
pmc_pass c4 30 --agents 16 --pois 256 --envs 8192 --steps-per-launch 30 --actions rng
pmc_pass c5 4 --agents 32 --pois 1024 --envs 16384 --steps-per-launch 4 --actions rng --comm-force-scale 0.5 --r-comm 0.1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/membw.hip -o $OUT/membw 2> $OUT/membw_build.err && {
  $OUT/membw > $OUT/membw_fill_copy.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_cal -- $OUT/membw > /dev/null 2> $OUT/pmc_cal.err
  python - <<'PY' > $OUT/pmc_WRITE_SIZE_calibration.txt
import csv, glob, collections
f = glob.glob('gpurun_out/profiles_r05_pmc/pmc_cal/**/*_counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r['Counter_Name'] == 'WRITE_SIZE':
        acc[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
known = 150 * 4096 * 676 * 16
for k, v in acc.items():
    per = sum(v) / len(v)
    if k.strip() in ('fill4', 'copy4', 'wave_blocks'):
        print("%-12s launches=%d  WRITE_SIZE=%.6g (KB units) -> %.4f GB per launch; known bytes written %.4f GB; calibration factor (known / counted) = %.4f"
              % (k.strip(), len(v), per, per * 1024 / 1e9, known / 1e9, known / (per * 1024)))
PY
  rm -rf $OUT/pmc_cal $OUT/membw $OUT/pmc_cal.err
}
python - <<'PY'
import json, re
out = 'gpurun_out/profiles_r05_pmc/'
def per_launch(name):
    for l in open(out + name):
        m = re.search(r'total=([0-9.e+]+)', l)
        if m and 'n=' in l:
            return float(m.group(1))
    return None
cal = 1.0
try:
    for l in open(out + 'pmc_WRITE_SIZE_calibration.txt'):
        if l.startswith('wave_blocks'):
            cal = float(l.rsplit('=', 1)[1])
except Exception:
    pass
src = ("rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE (separate passes of the rollout launches only: DCC_AUTOTUNE=0, --place-tries 0; KB units "
       "x1024; WRITE_SIZE calibrated on the known byte count of tools/membw.hip's wave_blocks store pattern, FETCH_SIZE doubled per "
       "MI355X_MICROARCH.md's gfx950 note), profiles/r05/pmc_%s_WRITE_SIZE.txt, pmc_%s_FETCH_SIZE.txt, pmc_WRITE_SIZE_calibration.txt (tools/pmc_traffic_round5.sh)")
for tag, N, M, E, T, acts, alg in (("c2", 8, 64, 4096, 150, "hbm", 11851), ("c4", 16, 256, 8192, 30, "rng", 87563 - 128), ("c5", 32, 1024, 16384, 4, "rng", 676363 - 256)):
    w, f = per_launch('pmc_%s_WRITE_SIZE.txt' % tag), per_launch('pmc_%s_FETCH_SIZE.txt' % tag)
    if w is None or f is None:
        continue
    wb, fb = w * 1024 * cal, f * 1024 * 2
    json.dump({"workload": {"n_agents": N, "n_pois": M, "envs": E, "steps_per_launch": T, "actions": acts},
               "write_bytes_per_launch": wb, "write_size_calibration_factor": cal, "fetch_bytes_per_launch_corrected_x2": fb,
               "traffic_bytes_per_launch": wb + fb, "traffic_bytes_per_env_step": (wb + fb) / (E * T),
               "algorithmic_bytes_per_launch": alg * E * T, "algorithmic_bytes_per_env_step": alg, "source": src % (tag, tag)},
              open(out + 'traffic_%s.json' % tag, 'w'), indent=1)
PY
ls -la $OUT
cat $OUT/traffic_c2.json
