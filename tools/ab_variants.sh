#!/bin/bash
# Same-box A/B of library variants under csrc/variants/*.so on the c2 headline and (optionally) the shard shapes:
# interleaved runs, roofline fraction + launch ms per variant.  usage: tools/ab_variants.sh [rounds] [extra bench args...]
cd ${GRAFT_REPO_ROOT:-/root/repo}
ROUNDS=${1:-2}; shift
for r in $(seq $ROUNDS); do for L in dynamic-coverage-control_amd/csrc/variants/*.so; do
  echo -n "$(basename $L) $@: "
  DCC_HIP_LIB=$PWD/$L python bench.py --no-c3 --no-cpu-baseline --steps 10 --warmup 3 "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(r['launch_ms_avg'],4), round(r['frac'],4), r.get('kernel_choice'))"
done; done
