#!/bin/bash
# How far does the REFERENCE's own Learner move when its initial parameters are scaled by (1 + 1e-7 N(0,1)) -- a 1-ulp-sized
# perturbation?  The noise floor any other implementation of the same arithmetic is compared against (container-only: imports
# /root/reference through tools/gen_golden_learner.py).  Output: profiles/r06/h256_fixture_sensitivity.txt
set -e
cd "$(dirname "$0")/.."
OUT=profiles/r06/h256_fixture_sensitivity.txt
T=$(mktemp -d)
cmp() { python - "$1" "$2" <<'PY'
import sys, numpy as np
A, B = np.load(sys.argv[1]), np.load(sys.argv[2])
for i in (1, 2, 3):
    w = dict(elem=0.0, l2=0.0)
    for k in A.files:
        if k.startswith("i%d/" % i) and k.endswith("#val") and float(A[k[:-4] + "#dmax"]) > 0:
            w["elem"] = max(w["elem"], float(np.abs(A[k].astype(np.float64) - B[k]).max() / float(A[k[:-4] + "#dmax"])))
            w["l2"] = max(w["l2"], abs(float(A[k[:-4] + "#dl2"]) - float(B[k[:-4] + "#dl2"])) / float(A[k[:-4] + "#dl2"]))
    print("   iteration %d: sampled elements %.1e of max|delta|, ||delta||_2 %.1e" % (i, w["elem"], w["l2"]))
k = max(int(f[1:].split("/")[0]) for f in A.files if f.startswith("r") and "/" in f)
print("   last rollout: rewards %.1e, value_preds %.1e of max" % tuple(
    float(np.abs(A["r%d/%s" % (k, n)] - B["r%d/%s" % (k, n)]).max() / np.abs(A["r%d/%s" % (k, n)]).max()) for n in ("rewards", "value_preds")))
PY
}
{
echo "# reference Learner, 8 UAV x 64 PoI, 2 envs, T = 40, hidden 256: unperturbed vs initial parameters * (1 + 1e-7 N(0,1))"
for ep in 15 5 3 2; do
  mkdir -p $T/base$ep
  python tools/gen_golden_learner.py 2 8 64 h256 ep$ep out=$T/base$ep > /dev/null 2>&1
  for seed in 1 2 3 4 5 6; do
    mkdir -p $T/p${ep}_$seed
    python tools/gen_golden_learner.py 2 8 64 h256 ep$ep pert=1e-7 seed=$seed out=$T/p${ep}_$seed > /dev/null 2>&1
    echo "ppo_epoch $ep, perturbation seed $seed"
    cmp $T/base$ep/learner_ref_e2_n8m64_h256.npz $T/p${ep}_$seed/learner_ref_e2_n8m64_h256.npz
  done
done
} | tee $OUT
rm -rf $T
