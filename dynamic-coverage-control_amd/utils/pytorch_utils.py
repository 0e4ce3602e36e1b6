"""Device plumbing used by train.py / learner.py (`ptu.set_gpu_mode`, `ptu.device`, tensor helpers).

Mirrors the names the reference's train.py and algos import (uav_dcc_control/utils/pytorch_utils.py:
121-180) so that an unchanged train.py runs.  Differences, on purpose:
  * `set_gpu_mode` does NOT export CUDA_VISIBLE_DEVICES (the reference sets it *and* uses
    cuda:<id>, which is inconsistent for id > 0 -- SURVEY.md Q10); it selects the HIP device.
  * one process per GPU: `init_distributed()` picks the device from LOCAL_RANK and brings up
    torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" on CPU).
"""
import os

import numpy as np
import torch

_use_gpu = False
_gpu_id = 0
device = torch.device("cpu")


def set_gpu_mode(mode, gpu_id=0):
    global _use_gpu, _gpu_id, device
    _gpu_id = int(gpu_id)
    _use_gpu = bool(mode)
    device = torch.device("cuda:%d" % _gpu_id if _use_gpu else "cpu")
    if _use_gpu:
        torch.cuda.set_device(device)


def gpu_enabled():
    return _use_gpu


TUNED_GEMMS_FILE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "config", "gemm_tunings_gfx950.csv")
_tuned_gemms = None


def use_tuned_gemms(path=None):
    """Library-GEMM selection for the policy MLPs: the plain fp32 GEMMs stay rocBLAS / hipBLASLt calls issued by torch, and
    for the shapes listed in config/gemm_tunings_gfx950.csv (the BASELINE sizes, produced by tools/tune_gemms.sh on an
    MI355X) torch's TunableOp dispatches the solution that measured fastest instead of the library heuristic's pick
    (c3: update GEMMs 5.0 -> 4.6-4.7 ms, +3.8 % per iteration).  LOOKUP ONLY -- nothing is tuned at run time, a shape that is
    not listed takes the default path, and torch ignores the file when its validators (torch / ROCm / rocBLAS / hipBLASLt
    versions, GPU arch) differ from this installation.  Returns the number of entries in use (0: default GEMMs).
    DCC_TUNED_GEMMS=0 or a user-managed PYTORCH_TUNABLEOP_ENABLED leave everything as it is."""
    global _tuned_gemms
    if _tuned_gemms is not None and path is None:
        return _tuned_gemms
    n = 0
    path = path or os.environ.get("DCC_TUNED_GEMMS_FILE") or TUNED_GEMMS_FILE      # the env override is for A/B runs
    if (_use_gpu and os.environ.get("DCC_TUNED_GEMMS", "1") != "0" and "PYTORCH_TUNABLEOP_ENABLED" not in os.environ
            and os.path.exists(path)):
        import torch.cuda.tunable as tun
        tun.enable(True)
        tun.tuning_enable(False)
        if tun.read_file(path):
            n = len(tun.get_results())
        else:
            tun.enable(False)
    _tuned_gemms = n
    return n


def testing_enabled():
    """DCC_TESTING=1: the ONE gate in front of every test seam of this package (tests/conftest.py and the tools that need them
    set it).  A production run cannot trip a seam by accident: without the gate the seams below refuse to act."""
    return os.environ.get("DCC_TESTING") == "1"


def require_testing(what):
    if not testing_enabled():
        raise RuntimeError("%s is a test seam of this package: it is honoured only with DCC_TESTING=1 in the environment" % what)


def test_hook(name, default=None):
    """Value of the TEST-ONLY environment switch `name` (DCC_DIST_SINGLE, DCC_DIST_BACKEND, DCC_BENCH_BACKEND, DCC_GLOO_VIA_HOST);
    set without DCC_TESTING=1 it is an error, not a silent change of behaviour."""
    v = os.environ.get(name)
    if v is None:
        return default
    require_testing("the environment switch %s" % name)
    return v


def init_distributed(backend=None):
    """One process per GPU.  Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment
    (torch.distributed.run); no-op when WORLD_SIZE is 1.  Returns (rank, world_size)."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1 and not single_rank_group():
        return 0, 1
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world == 1:
        os.environ.setdefault("MASTER_PORT", "29571")
    if backend is None:   # DCC_DIST_BACKEND=gloo is a test hook: the ranks may then share one GPU (gloo moves CUDA tensors)
        backend = test_hook("DCC_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        set_gpu_mode(True, int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(backend, rank=rank, world_size=world, device_id=device)
    else:
        if torch.cuda.is_available():
            set_gpu_mode(True, int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world


def single_rank_group():
    """DCC_DIST_SINGLE=1 (a test hook for boxes with ONE GPU): bring up the process group with a single rank and take every
    distributed code path anyway -- over backend "nccl" that runs communicator creation and each collective call site
    (all-reduce sync / async, broadcast, gather_object, barrier) through RCCL on real hardware, which is where a tensor left
    on the host or an operation the backend lacks would show."""
    return test_hook("DCC_DIST_SINGLE") == "1"


def dist_active():
    """the process group is up and the distributed code paths are to be taken"""
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or single_rank_group())


class _Done(object):
    """Handle of a collective that has already completed (the host-staged gloo path below)."""

    def wait(self):
        return True


def _stage_through_host(t):
    """True when a collective on device tensor `t` is to be staged through the host by THIS code.
    RCCL (backend "nccl", the product path: one process per GPU) reduces device buffers in place.  gloo has no device transport:
    torch's ProcessGroupGloo moves a CUDA tensor through pinned staging buffers it allocates per call and SDMA copies on streams
    of its own.  With EIGHT processes sharing one GPU (the gloo test hook: DCC_DIST_BACKEND / DCC_BENCH_BACKEND = gloo) that path
    makes a rank die with HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION -- in whatever kernel is running, stock torch kernels included --
    in 42 of 144 jobs; with the copies made here as plain `.cpu()` / `copy_` on the current stream 0 of 34, and 0 of 54 with
    HSA_ENABLE_SDMA=0 (profiles/r05/world8_ab.txt).  So on gloo the tensors are staged here; DCC_GLOO_VIA_HOST=0 restores torch's
    path (the A/B knob of tools/world8_ab.py)."""
    import torch.distributed as dist
    return t.is_cuda and dist.get_backend() == "gloo" and test_hook("DCC_GLOO_VIA_HOST", "1") != "0"


def all_reduce(t, async_op=False):
    """dist.all_reduce(t) (sum, in place).  async_op: returns a handle with .wait()."""
    import torch.distributed as dist
    if _stage_through_host(t):
        h = t.cpu()
        dist.all_reduce(h)
        t.copy_(h)
        return _Done() if async_op else None
    return dist.all_reduce(t, async_op=async_op)


def broadcast(t, src=0):
    import torch.distributed as dist
    if _stage_through_host(t):
        h = t.cpu()
        dist.broadcast(h, src)
        t.copy_(h)
        return
    dist.broadcast(t, src)


def world_size():
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def from_numpy(*args, **kwargs):
    return torch.from_numpy(*args, **kwargs).float().to(device)


def to_tensor(x, dtype=torch.float32):
    """numpy / tensor -> tensor on `device` (no copy when it already lives there)."""
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    return x.to(device=device, dtype=dtype)


def get_numpy(tensor):
    return tensor.detach().to("cpu").numpy()


def zeros(*sizes, **kwargs):
    return torch.zeros(*sizes, **kwargs, device=device)


def ones(*sizes, **kwargs):
    return torch.ones(*sizes, **kwargs, device=device)


def eye(*sizes, **kwargs):
    return torch.eye(*sizes, **kwargs, device=device)


def randn(*args, **kwargs):
    return torch.randn(*args, **kwargs, device=device)


def zeros_like(*args, **kwargs):
    return torch.zeros_like(*args, **kwargs).to(device)


def ones_like(*args, **kwargs):
    return torch.ones_like(*args, **kwargs).to(device)
