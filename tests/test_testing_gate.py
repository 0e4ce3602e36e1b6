"""The ONE gate in front of the package's test seams: injected action noise, injected mini-batch permutations,
the gloo / single-rank process-group hooks and bench.py's lost-rank flag act only with DCC_TESTING=1 in the environment (this
suite's conftest sets it); without it they are errors, not silent changes of behaviour."""
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT


def test_seams_refuse_without_the_gate(monkeypatch):
    import utils.pytorch_utils as ptu
    from algos.algo_utils import distributions
    monkeypatch.delenv("DCC_TESTING")
    assert not ptu.testing_enabled()
    with pytest.raises(RuntimeError, match="DCC_TESTING=1"):
        distributions.set_noise_source(lambda shape, dtype, device: torch.zeros(shape))
    assert distributions.set_noise_source(None) is None                  # restoring torch.randn is always allowed
    for name in ("DCC_DIST_SINGLE", "DCC_DIST_BACKEND", "DCC_GLOO_VIA_HOST"):
        assert ptu.test_hook(name, "dflt") == "dflt"                     # unset: the default, no error
        monkeypatch.setenv(name, "1")
        with pytest.raises(RuntimeError, match=name):
            ptu.test_hook(name)
        monkeypatch.delenv(name)
    monkeypatch.setenv("DCC_DIST_SINGLE", "1")
    with pytest.raises(RuntimeError):
        ptu.single_rank_group()


def test_seams_act_with_the_gate(monkeypatch):
    import utils.pytorch_utils as ptu
    from algos.algo_utils import distributions
    assert ptu.testing_enabled()
    src = lambda shape, dtype, device: torch.full(shape, 0.25, dtype=dtype, device=device)
    prev = distributions.set_noise_source(src)
    try:
        assert float(distributions.standard_normal((2, 3), torch.float32, torch.device("cpu")).mean()) == 0.25
    finally:
        distributions.set_noise_source(prev)
    monkeypatch.setenv("DCC_DIST_SINGLE", "1")
    assert ptu.single_rank_group()


def test_injected_permutations_need_the_gate(monkeypatch):
    from test_mappo_env_golden import Box, _fill, _perms, _policy, case, make_cfg
    from buffer.shared_buffer import SharedReplayBuffer
    c = case("small_mb2")
    cfg = make_cfg(c, structured_input=False, compact_obs=False)
    pol, tr = _policy(c, cfg, gpu=False)
    buf = SharedReplayBuffer(cfg, Box(c.D), Box(c.S), Box(c.A), n_pois=c.M)
    _fill(c, buf)
    tr.prep_training()
    tr.minibatch_perms = _perms(c)
    monkeypatch.delenv("DCC_TESTING")
    with pytest.raises(RuntimeError, match="minibatch_perms"):
        tr.train(buf, update_actor=True)


def test_bench_refuses_its_test_hooks_without_the_gate():
    env = {k: v for k, v in os.environ.items() if k != "DCC_TESTING"}
    for extra_env, args in ((dict(DCC_BENCH_BACKEND="gloo"), []), ({}, ["--test-kill-rank-at-leg", "0:c4"])):
        r = subprocess.run([sys.executable, "bench.py", "--steps", "1", "--warmup", "0"] + args, cwd=ROOT, capture_output=True, text=True,
                           timeout=300, env=dict(env, **extra_env))
        # (without a GPU bench.py stops even earlier, with its own message: both are refusals, neither runs a seam)
        assert r.returncode != 0 and ("DCC_TESTING=1" in r.stderr or "needs a GPU" in r.stderr), r.stderr[-500:]
