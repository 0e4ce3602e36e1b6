#!/bin/bash
# Run on the GPU box: the checks beyond the parity suites for round 5's binary -- host-side UBSan through the C-ABI GPU suites,
# graph-vs-eager soak of the rollout (now with dcc_env_step_features inside the captured graph), both differential fuzzers.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/harden_r05
bash tools/ubsan_host_build.sh > gpurun_out/harden_r05/ubsan_host.txt 2>&1; tail -3 gpurun_out/ubsan.log >> gpurun_out/harden_r05/ubsan_host.txt
python tools/graph_rollout_soak.py 150 4096 8 64 150 > gpurun_out/harden_r05/graph_rollout_soak_c3.txt 2>&1
python tools/graph_rollout_soak.py 400 64 4 20 50 > gpurun_out/harden_r05/graph_rollout_soak_small.txt 2>&1
python tools/fuzz_env_parity.py 240 555 > gpurun_out/harden_r05/fuzz_env_parity_2.txt 2>&1
python tools/fuzz_mlp_kernels.py 120 > gpurun_out/harden_r05/fuzz_mlp_kernels.txt 2>&1
tail -2 gpurun_out/harden_r05/*.txt
