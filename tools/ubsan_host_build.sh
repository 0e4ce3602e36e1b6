#!/bin/bash
# Host-side UndefinedBehaviorSanitizer build of libdcc_hip.so (the launcher / argument / workspace code around the kernels;
# device code cannot be instrumented for gfx950) and the GPU suites that go through the C-ABI against it.
# Run on the GPU box: bash tools/ubsan_host_build.sh      -> gpurun_out/ubsan.log
# Notes from round 2: (1) AddressSanitizer is not usable for the in-process library here -- the HIP runtime of this image does
# not start under ASan's allocator (out-of-memory in libamdhip64's address-space reservation; SEGV with
# allocator_may_return_null=1) and the image has no ASan-enabled ROCm runtime; the C restatement and the _cpu twins ARE run
# under ASan + UBSan (tests/test_sanitizers.py).  (2) -fsanitize=function / vptr / object-size trap inside HIP's kernel-launch
# stubs (SIGILL at the first launch, without a report), so the check list below names the others explicitly.
set -e
cd "$(dirname "$0")/.."
RT=$(dirname $(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.ubsan_standalone-x86_64.so))
mkdir -p dynamic-coverage-control_amd/csrc/variants gpurun_out
( cd dynamic-coverage-control_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -ffp-contract=off \
    -fno-fast-math -fvisibility=hidden -shared-libsan \
    -fsanitize=signed-integer-overflow,shift,integer-divide-by-zero,bounds,null,alignment,bool,enum,float-cast-overflow,return,vla-bound,pointer-overflow \
    -I../../include -DDCC_BUILDING=1 -o variants/libdcc_hip_ubsan.so dcc_env.hip dcc_gae.hip dcc_mlp.hip dcc_optim.hip 2>&1 | grep -v "warning: ignoring\|^$" || true )
export UBSAN_OPTIONS=print_stacktrace=1 DCC_HIP_LIB=$PWD/dynamic-coverage-control_amd/csrc/variants/libdcc_hip_ubsan.so
export LD_LIBRARY_PATH=$RT:$LD_LIBRARY_PATH
python -m pytest tests/test_env_hip_parity.py tests/test_mlp_fused_hip.py tests/test_gae_hip.py tests/test_structured_input.py \
    tests/test_abi_exports.py tests/test_integration_stub.py tests/test_learner_hip.py tests/test_full_size_gpu.py -q -m gpu > gpurun_out/ubsan.log 2>&1 || true
echo "UBSan reports: $(grep -c 'runtime error' gpurun_out/ubsan.log)"; tail -1 gpurun_out/ubsan.log
