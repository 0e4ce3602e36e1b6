#!/bin/bash
# Same-box A/B of library variants under csrc/variants/*.so: interleaved runs of the default bench.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2 3; do for L in dynamic-coverage-control_amd/csrc/variants/*.so; do
  echo -n "$(basename $L) $@: "
  DCC_HIP_LIB=$PWD/$L python bench.py --no-cpu-baseline --steps 7500 --warmup 750 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2), round(d['roofline']['frac'],3))"
done; done
