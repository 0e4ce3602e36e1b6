#!/bin/bash
# Same-box interleaved A/B of csrc/variants/*.so on the c4 and c5 per-GPU shards (placement-probed output buffers).
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2; do for L in dynamic-coverage-control_amd/csrc/variants/*.so; do
  for S in "c4:--agents 16 --pois 256 --envs 1024 --steps 5 --warmup 2 --launches-per-step 8" "c5:--agents 32 --pois 1024 --envs 2048 --comm-force-scale 0.5 --r-comm 0.1 --steps-per-launch 50 --steps 3 --warmup 1 --launches-per-step 2"; do
    NAME=${S%%:*}; ARGS=${S#*:}
    echo -n "$(basename $L) $NAME: "
    DCC_HIP_LIB=$PWD/$L python bench.py --no-c3 --no-cpu-baseline $ARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(r['launch_ms_avg'],4), round(r['frac'],4))"
  done
done; done
