"""Boundary B1-B3: the reference's entry point and orchestrator find in this package everything they import, open,
call and touch.

WHAT they need is a list of identifiers -- modules, functions and their argument shapes, YAML paths and their merge order, cfg
attributes read / assigned, vec-env / buffer / policy / trainer attributes, dictionary keys -- kept as DATA in
tests/golden/b1_contract.json.  That file is extracted from the reference's two files by tools/check_b1_contract.py in the build
container (syntax-tree walk, names only); here, and on the GPU box where the reference does not exist, the package is checked
against it: statically on the CPU (modules, signatures, YAML keys), and live on the GPU by tests/_b1_driver.py, which builds a
DictConfig layer by layer from the contract's paths, goes through the contract's calls and runs the learner.  `omegaconf` is not
installed in this image, so a test-only stand-in with the same container API (tests/_standin/omegaconf) is used unless the real
package is importable."""
import importlib
import inspect
import json
import os
import subprocess
import sys

import pytest
import yaml

from conftest import GOLDEN, PKG, ROOT

CONTRACT = os.path.join(GOLDEN, "b1_contract.json")
DRIVER = os.path.join(ROOT, "tests", "_b1_driver.py")


def _contract():
    with open(CONTRACT) as f:
        return json.load(f)


def _env():
    env = dict(os.environ)
    extra = [PKG]
    try:
        import omegaconf  # noqa: F401  (the real package wins when present)
    except ImportError:
        extra.append(os.path.join(ROOT, "tests", "_standin"))
    env["PYTHONPATH"] = os.pathsep.join(extra + [env.get("PYTHONPATH", "")])
    return env


def _accepts(fn, n_positional, keywords):
    try:
        inspect.signature(fn).bind(*([None] * n_positional), **{k: None for k in keywords})
        return True
    except TypeError:
        return False


def test_contract_file_is_names_only():
    """The fixture carries identifiers and argument counts, nothing else: every leaf is a short token, a path or a small integer."""
    def leaves(x):
        if isinstance(x, dict):
            for k, v in x.items():
                yield k
                yield from leaves(v)
        elif isinstance(x, list):
            for v in x:
                yield from leaves(v)
        else:
            yield x
    c = _contract()
    c.pop("_about")
    for leaf in leaves(c):
        assert isinstance(leaf, (str, int))
        if isinstance(leaf, str):
            assert len(leaf) < 48 and "\n" not in leaf and " " not in leaf, leaf


def test_package_answers_the_static_part_of_the_contract():
    """CPU: modules and names importable, call shapes bindable, Learner's methods with the parameter names the reference's
    own call sites use, every cfg attribute that is read present in the merged YAML layers, in the contract's merge order."""
    c = _contract()
    entry, orch = c["train_py"], c["learner_py"]
    for section in (entry, orch):
        for mod_name, names in section["imports"].items():
            mod = importlib.import_module(mod_name)
            for n in names:
                assert hasattr(mod, n) or importlib.util.find_spec(mod_name + "." + n), (mod_name, n)
    for callee, shapes in list(entry["calls"].items()) + list(orch["calls"].items()):
        mod_name, _, fn_name = callee.rpartition(".")
        if mod_name in ("vec_env", "buffer", "policy", "trainer"):
            continue                                       # methods of live objects: the GPU test
        fn = getattr(importlib.import_module(mod_name), fn_name)
        for shape in shapes:
            assert _accepts(fn, shape["positional"], shape["keywords"]), callee
    from learner import Learner
    for name, params in orch["learner_methods"].items():
        have = [p for p in inspect.signature(getattr(Learner, name)).parameters if p != "self"]
        assert have[:len(params)] == params, (name, have)
    from algos.mappo import MAPPOPolicy, MAPPOTrainer
    from buffer.shared_buffer import SharedReplayBuffer
    for cls, key in ((MAPPOPolicy, "policy_attributes"), (SharedReplayBuffer, "buffer_attributes")):
        for attr in orch[key]:
            member = getattr(cls, attr, None)
            if callable(member):                            # methods are visible on the class; tensors only on an instance
                for shape in orch["calls"].get(("policy." if cls is MAPPOPolicy else "buffer.") + attr, []):
                    assert _accepts(member, shape["positional"] + 1, shape["keywords"]), attr
    for attr in ("prep_rollout", "prep_training", "train", "save_model", "load_model"):
        assert attr in orch["trainer_attributes"] and callable(getattr(MAPPOTrainer, attr))
    merged = {}
    for path in entry["yaml_paths_in_merge_order"]:
        with open(os.path.join(PKG, path)) as f:
            merged.update(yaml.safe_load(f))                # later wins
    assert sorted(entry["yaml_paths"]) == sorted(entry["yaml_paths_in_merge_order"])
    for section in (entry, orch):
        for key, mode in section["cfg_attributes"].items():
            if mode == "r":
                assert key in merged, "cfg.%s is read by the reference but no YAML layer defines it" % key
    # the two layers that disagree do so the way the reference's order resolves them (the algo file's eval / render
    # thread counts override the env file's)
    with open(os.path.join(PKG, entry["yaml_paths_in_merge_order"][1])) as f:
        algo = yaml.safe_load(f)
    assert merged["n_eval_rollout_threads"] == algo["n_eval_rollout_threads"]


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
def test_live_package_answers_the_whole_contract(tmp_path):
    """GPU: the shipped YAMLs (4 UAV x 20 PoI, 16 envs, 150-step rollouts) as a layered DictConfig through every call, attribute
    and key of the contract, then two training iterations; the checkpoint lands at <main_save_path>/<save_name>/<expt>/models_N.pt."""
    r = subprocess.run([sys.executable, DRIVER, CONTRACT, "0", str(tmp_path) + "/", "n_iters=2", "save_interval=2", "ppo_epoch=3"],
                       cwd=PKG, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    report = json.loads([l for l in r.stdout.splitlines() if l.startswith("B1_REPORT ")][0][len("B1_REPORT "):])
    assert report["missing"] == []
    assert "iter: 2" in r.stdout and "model saved" in r.stdout
    out = report["output_path"]
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
