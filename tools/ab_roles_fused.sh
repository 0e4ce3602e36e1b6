#!/bin/bash
# Same-box A/B of the role-specialised and the fused c2 kernel (interleaved, HIP-event launch averages).
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2; do for v in 0 1; do
  echo -n "DCC_NO_ROLES=$v: "
  DCC_NO_ROLES=$v python bench.py --steps 10 --warmup 3 --no-c3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms  frac %.3f' % (d['roofline']['launch_ms_avg'], d['roofline']['frac']))"
done; done
