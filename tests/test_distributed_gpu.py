"""-m gpu: a 2-rank MAPPO job on the GPU through torch.distributed.run (both ranks share the one device over the gloo
test hook, DCC_DIST_BACKEND=gloo): the shipped config's multi-rank path runs end to end and the replicas stay identical."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mini_batches", [1, 2])
def test_two_rank_gpu_training_keeps_replicas_identical(mini_batches):
    """mini_batches = 2: the reference's row mini-batches on the shipped state-only storage, each rank over its own env shard."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29540 + mini_batches), os.path.join(ROOT, "tests", "_dist_gpu_worker.py")]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, DCC_DIST_BACKEND="gloo", DCC_TEST_MINI_BATCH=str(mini_batches)))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "DIST_GPU_OK" in r.stdout


def test_train_py_as_a_two_rank_job(tmp_path):
    """The launcher under torch.distributed.run, one process per rank (the ranks share the one GPU over the gloo test hook):
    2 iterations, the checkpoint written collectively; test_two_gpus_over_rccl runs the same over RCCL where 2 GPUs exist."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29543", "train.py", "0", "n_iters=2", "n_rollout_threads=64", "n_eval_rollout_threads=0", "max_ep_len=10",
           "ppo_epoch=2", "algo_hidden_size=32", "save_interval=2", "main_save_path=%s/" % tmp_path]
    r = subprocess.run(cmd, cwd=os.path.join(ROOT, "dynamic-coverage-control_amd"), capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, DCC_DIST_BACKEND="gloo"))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "iter: 2" in r.stdout and "model saved" in r.stdout and "2 GPUs" in r.stdout


def test_single_rank_group_over_rccl():
    """What a 1-GPU box can run of the RCCL path: a ONE-rank process group over backend nccl (DCC_DIST_SINGLE=1) through every
    collective call site of the learner and of bench.py -- a host tensor handed to a collective or an operation RCCL lacks
    fails here, not first on the 8-GPU node."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DCC_BENCH_BACKEND", "DCC_DIST_BACKEND")}
    env.update(DCC_DIST_SINGLE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_single_worker.py")], cwd=ROOT, capture_output=True, text=True,
                       timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "RCCL_SINGLE_OK" in r.stdout
    env["MASTER_PORT"] = "29548"
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--launches-per-step", "4",
                        "--c3-iters", "1", "--ppo-epoch", "2", "--envs", "1024", "--no-cpu-baseline"], cwd=ROOT, capture_output=True,
                       text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 1 and "error" not in d["c3"] and "error" not in d["c4"] and "error" not in d["c5"], (d["c3"], d["c4"], d["c5"])
    assert d["c3"]["grad_allreduce"] == "rccl x1" and all(v == v for v in d["c3"]["train_info"].values())


@pytest.mark.skipif(__import__("torch").cuda.device_count() < 2, reason="needs two GPUs (the gpurun box has one)")
def test_two_gpus_over_rccl():
    """Runs wherever >= 2 GPUs are visible: bench.py --gpus 2 launches itself, the ranks rendezvous over RCCL
    (backend nccl), the env shards scale weakly and the bounded c3 leg all-reduces the flat gradients over xGMI."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DCC_BENCH_BACKEND",
                                                             "DCC_DIST_BACKEND")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--launches-per-step", "4",
                        "--c3-iters", "1", "--ppo-epoch", "2", "--envs", "1024"], cwd=ROOT, capture_output=True, text=True,
                       timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["config"]["global_envs"] == 2048 and "error" not in d["c3"], d["c3"]
    assert d["c3"]["grad_allreduce"] == "rccl x2" and all(v == v for v in d["c3"]["train_info"].values())
    # the line proves the job it ran as: two distinct devices seen over RCCL, an all-reduce of ones == 2, the strong c2 figure
    rc = d["rccl"]
    assert rc["backend"] == "rccl" and rc["world_size"] == 2 and rc["allreduce_ok"] and rc["distinct_devices"] == 2, rc
    assert "error" not in d["c2_strong"] and d["c2_strong"]["envs_per_gpu"] == 2048
    # the replica-identity worker (parameters / ValueNorm digests equal on both ranks after 3 iterations, per-rank resume) over RCCL
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1"]
    r = subprocess.run(launcher + ["--master-port", "29551", os.path.join(ROOT, "tests", "_dist_gpu_worker.py")], cwd=ROOT,
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "DIST_GPU_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    # and the launcher itself: train.py for 2 iterations as a 2-rank job (one process per GPU), checkpoint written collectively
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        r = subprocess.run(launcher + ["--master-port", "29552", "train.py", "0", "n_iters=2", "n_rollout_threads=64",
                                       "n_eval_rollout_threads=0", "max_ep_len=10", "ppo_epoch=2", "algo_hidden_size=32",
                                       "save_interval=2", "main_save_path=%s/" % tmp],
                           cwd=os.path.join(ROOT, "dynamic-coverage-control_amd"), capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        assert "iter: 2" in r.stdout and "model saved" in r.stdout and "2 GPUs" in r.stdout
