"""-m gpu: `bench.py` honours the driver's contract -- one JSON line with the agreed keys, the roofline object measured
with HIP events on the launch stream, a bounded cpu_baseline leg at N = 1 -- for a short run; and two ranks sharing the GPU
over the gloo test hook report the job-wide aggregate (weak scaling: per-GPU work fixed)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline"}


def _run(args, env=None, launcher=None):
    cmd = (launcher or [sys.executable]) + ["bench.py"] + args
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]          # exactly ONE JSON line
    return json.loads(lines[0])


def test_single_gpu_line_has_the_contract_keys():
    d = _run(["--gpus", "1", "--steps", "450", "--warmup", "150"])
    assert KEYS <= set(d) and "cpu_baseline" in d
    assert d["metric"] == "agent_env_steps_per_sec" and d["unit"] == "agent-env-steps/s" and d["n_gpus"] == 1
    assert d["steps"] == 450 and d["warmup"] == 150 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f64"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 4096 * 8 * 450 / (d["ms_per_step"] * 450 / 1e3)) / d["value"] < 1e-6
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert 0.3 < r["frac"] < 1.0 and r["traffic"] and r["bytes_per_env_step"] == 11851
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 1e5 and "sample" in c
    assert d["value"] > 50e6                          # the north-star floor, by a wide margin


def test_two_ranks_report_the_aggregate():
    """torch.distributed.run with 2 ranks on the one GPU (gloo rendezvous hook): n_gpus = 2, twice the envs, no cpu_baseline."""
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                "127.0.0.1", "--master-port", "29533"]
    d = _run(["--gpus", "2", "--steps", "300", "--warmup", "150", "--envs", "1024"], env={"DCC_BENCH_BACKEND": "gloo"},
             launcher=launcher)
    assert d["n_gpus"] == 2 and d["config"]["global_envs"] == 2048 and d["config"]["envs_per_gpu"] == 1024
    assert "cpu_baseline" not in d and d["scaling"] == "weak"
    assert abs(d["value"] - 2 * 1024 * 8 * 300 / (d["ms_per_step"] * 300 / 1e3)) / d["value"] < 1e-6
