#!/bin/bash
# Same-box interleaved A/B of csrc/variants/*.so on c2 and the c4 per-GPU shard (create-time kernel choice off: kernels compared directly).
cd ${GRAFT_REPO_ROOT:-/root/repo}
export DCC_AUTOTUNE=0
for r in 1 2 3; do for L in dynamic-coverage-control_amd/csrc/variants/*.so; do
  for S in "c2:--steps 10 --warmup 3" "c4:--agents 16 --pois 256 --envs 1024 --steps 5 --warmup 2 --launches-per-step 8"; do
    NAME=${S%%:*}; ARGS=${S#*:}
    echo -n "$(basename $L) $NAME: "
    DCC_HIP_LIB=$PWD/$L python bench.py --no-c3 --no-cpu-baseline $ARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(r['launch_ms_avg'],4), round(r['frac'],4))"
  done
done; done
