"""-m gpu: boundary B1 -- the reference's entry point drives this package unchanged.

uav_dcc_control/train.py:12-29 does, from the package directory: load the three YAML files with OmegaConf, merge
them (env, algo, expt -- later wins), `ptu.set_gpu_mode(cuda available, gpu_id=argv[1])`, `torch.set_num_threads`,
`os.makedirs(cfg.main_save_path)`, ASSIGN `cfg.log_wandb = False; cfg.save_model = True` on the merged DictConfig,
then `Learner(cfg).train()`.  The driver below restates that call sequence (the reference file itself cannot travel to
the GPU box) against a DictConfig-shaped object; `omegaconf` is not installed in this image, so a test-only stand-in
with the same container API (tests/_standin/omegaconf) is put on PYTHONPATH.  Where the real package exists it is
used instead."""
import os
import subprocess
import sys

import pytest

from conftest import PKG, ROOT

DRIVER = r'''
import os, sys, torch
from omegaconf import OmegaConf
import utils.pytorch_utils as ptu
from learner import Learner
env_cfg = OmegaConf.load("./config/env_config/dcc.yaml")
ptu.set_gpu_mode(torch.cuda.is_available(), gpu_id=int(sys.argv[1]))
algo_cfg = OmegaConf.load("./config/algo_config/mappo.yaml")
expt_cfg = OmegaConf.load("./config/expt.yaml")
cfg = OmegaConf.merge(env_cfg, algo_cfg, expt_cfg)
assert OmegaConf.is_config(cfg) and not isinstance(cfg, dict)
torch.set_num_threads(min(cfg.n_training_threads, os.cpu_count() or 1))
cfg.main_save_path = sys.argv[2]
os.makedirs(cfg.main_save_path, exist_ok=True)
cfg.log_wandb = False
cfg.save_model = True
for kv in sys.argv[3:]:                      # test sizing only: a short run
    k, v = kv.split("=")
    cfg[k] = type(cfg[k])(v) if not isinstance(cfg[k], bool) else v == "True"
learner = Learner(cfg)
assert cfg.save_model is True and cfg.num_pois == learner.train_envs.n_pois     # the caller's object is still usable
learner.train()
print("OUTPUT_PATH", learner.output_path)
'''


def _env():
    env = dict(os.environ)
    extra = [PKG]
    try:
        import omegaconf  # noqa: F401  (the real package wins when present)
    except ImportError:
        extra.append(os.path.join(ROOT, "tests", "_standin"))
    env["PYTHONPATH"] = os.pathsep.join(extra + [env.get("PYTHONPATH", "")])
    return env


def test_standin_has_the_container_semantics_the_entry_point_relies_on(tmp_path):
    """CPU: attribute get/set, later-wins merge, `5e-4` read as a float (OmegaConf / YAML 1.2 behaviour; PyYAML alone
    returns a string), is_config / to_container -- and Learner's cfg conversion accepts the object."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "_standin"))
    try:
        from omegaconf import OmegaConf
        a, b = tmp_path / "a.yaml", tmp_path / "b.yaml"
        a.write_text("x: 1\nlr: 5e-4\nname: env\n")
        b.write_text("x: 2\nflag: true\n")
        cfg = OmegaConf.merge(OmegaConf.load(str(a)), OmegaConf.load(str(b)))
        assert cfg.x == 2 and cfg.lr == 5e-4 and isinstance(cfg.lr, float) and cfg.flag is True and cfg.name == "env"
        cfg.save_model = True
        assert cfg.save_model is True and OmegaConf.is_config(cfg) and not isinstance(cfg, dict)
        with pytest.raises(AttributeError):
            cfg.missing
        from learner import _to_namespace
        ns = _to_namespace(cfg)
        assert ns.x == 2 and ns.save_model is True and ns.lr == 5e-4
        ns.x = 3
        assert cfg.x == 2                          # a copy: Learner never writes into the caller's config
    finally:
        sys.path.pop(0)
        sys.modules.pop("omegaconf", None)


@pytest.mark.gpu
def test_reference_entry_sequence_with_a_dictconfig(tmp_path):
    """The shipped YAMLs (4 UAV x 20 PoI, 16 envs, 150-step rollouts) through the reference's call sequence, two
    iterations; the checkpoint lands where the reference puts it (<main_save_path>/<save_name>/<expt>/models_N.pt)."""
    r = subprocess.run([sys.executable, "-c", DRIVER, "0", str(tmp_path) + "/", "n_iters=2", "save_interval=2", "ppo_epoch=3"],
                       cwd=PKG, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "iter: 2" in r.stdout and "model saved" in r.stdout
    out = [l.split()[1] for l in r.stdout.splitlines() if l.startswith("OUTPUT_PATH")][0]
    assert os.path.exists(os.path.join(out, "models_2.pt", "agent.pkl")) and os.path.exists(os.path.join(out, "config.json"))


@pytest.mark.gpu
def test_baseline_config1_through_the_launcher(tmp_path):
    """BASELINE configs[0] (c1): 4 UAVs x 16 PoIs x 1 env, env.step + MAPPO via train.py -- plumbing on the GPU path
    (one env = one wavefront), full 150-step rollouts, two iterations."""
    r = subprocess.run([sys.executable, "train.py", "0", "num_agents=4", "num_pois=16", "n_rollout_threads=1",
                        "n_eval_rollout_threads=1", "n_iters=2", "ppo_epoch=5", "save_interval=2",
                        "main_save_path=%s/" % tmp_path], cwd=PKG, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "1 envs on this GPU" in r.stdout and "iter: 2" in r.stdout and "model saved" in r.stdout
