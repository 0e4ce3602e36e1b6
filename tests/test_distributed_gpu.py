"""-m gpu: a 2-rank MAPPO job on the GPU through torch.distributed.run (both ranks share the one device over the gloo
test hook, DCC_DIST_BACKEND=gloo): the shipped config's multi-rank path runs end to end and the replicas stay identical."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_two_rank_gpu_training_keeps_replicas_identical():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "tests", "_dist_gpu_worker.py")]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, DCC_DIST_BACKEND="gloo"))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "DIST_GPU_OK" in r.stdout
