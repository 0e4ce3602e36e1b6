"""Boundary B1-B3 exercised from the NAME contract (tests/golden/b1_contract.json, extracted from the reference's entry point and
orchestrator by tools/check_b1_contract.py in the build container): everything this script imports, opens, calls and touches is
looked up in that file, so it holds no statement of the reference -- only what the package has to answer to.

    python tests/_b1_driver.py <contract.json> <gpu_id> <save_root> [key=value ...]      (cwd = the package directory)

Prints one JSON object: {"missing": [...], "output_path": ..., "iters_logged": n}.  `missing` lists every identifier of the
contract the live objects did not provide."""
import importlib
import inspect
import json
import os
import sys


def accepts(fn, n_positional, keywords):
    """Can `fn` be called with that many positional arguments plus those keyword names?"""
    try:
        inspect.signature(fn).bind(*([None] * n_positional), **{k: None for k in keywords})
        return True
    except TypeError:
        return False


def main(argv):
    contract_path, gpu_id, save_root, overrides = argv[0], int(argv[1]), argv[2], argv[3:]
    with open(contract_path) as f:
        contract = json.load(f)
    entry, orch = contract["train_py"], contract["learner_py"]
    missing = []

    import torch
    from omegaconf import OmegaConf

    # -- what the entry point imports -------------------------------------------------------------------------------------
    modules = {}
    for mod_name, names in entry["imports"].items():
        modules[mod_name] = importlib.import_module(mod_name)
        missing += ["%s.%s" % (mod_name, n) for n in names if not hasattr(modules[mod_name], n)]

    # -- the layered config: one DictConfig per YAML path, merged in the contract's order (later wins) ----------------------
    layers = []
    for path in entry["yaml_paths_in_merge_order"]:
        if not os.path.exists(path):
            missing.append("file " + path)
            continue
        layers.append(OmegaConf.load(path))
    cfg = OmegaConf.merge(*layers)
    assert OmegaConf.is_config(cfg) and not isinstance(cfg, dict)

    # -- every first-party call of the entry point, with the argument shape it uses ------------------------------------------
    for callee, shapes in entry["calls"].items():
        mod_name, _, fn_name = callee.rpartition(".")
        fn = getattr(importlib.import_module(mod_name), fn_name, None)
        if fn is None or not all(accepts(fn, s["positional"], s["keywords"]) for s in shapes):
            missing.append("call " + callee)
    select = entry["calls"]["utils.pytorch_utils.set_gpu_mode"][0]
    importlib.import_module("utils.pytorch_utils").set_gpu_mode(torch.cuda.is_available(), **{select["keywords"][0]: gpu_id})

    # -- cfg attributes the entry point reads must come out of the YAML layers; the ones it assigns must be assignable -----
    for key, mode in entry["cfg_attributes"].items():
        if "r" in mode and key not in cfg:
            missing.append("cfg." + key)
    torch.set_num_threads(max(1, min(int(cfg.n_training_threads), os.cpu_count() or 1)))
    cfg.main_save_path = save_root
    os.makedirs(cfg.main_save_path, exist_ok=True)
    assigned = {"log_wandb": False, "save_model": True}        # the values the reference's entry point forces
    for key, mode in entry["cfg_attributes"].items():
        if "w" in mode:
            setattr(cfg, key, assigned[key])
    for kv in overrides:                                         # test sizing only
        key, _, val = kv.partition("=")
        old = cfg[key]
        cfg[key] = (val == "True") if isinstance(old, bool) else type(old)(val)

    # -- cfg attributes the orchestrator reads (the ones it writes itself excluded) --------------------------------------------
    for key, mode in orch["cfg_attributes"].items():
        if mode == "r" and key not in cfg:
            missing.append("cfg." + key)

    learner_cls = getattr(modules["learner"], "Learner")
    for name, params in orch["learner_methods"].items():
        fn = getattr(learner_cls, name, None)
        if fn is None:
            missing.append("Learner." + name)
            continue
        have = [p for p in inspect.signature(fn).parameters if p != "self"]
        if have[:len(params)] != params:
            missing.append("Learner.%s(%s)" % (name, ", ".join(params)))

    learner = learner_cls(cfg)
    assert cfg.save_model is True                                 # the caller's object is still its own

    # -- live objects: vec-env, buffer, policy, trainer ------------------------------------------------------------------------
    roles = {"vec_env": learner.train_envs, "buffer": learner.rl_buffer, "trainer": learner.trainer, "policy": learner.trainer.policy}
    for role, key in (("vec_env", "vec_env_attributes"), ("buffer", "buffer_attributes"), ("policy", "policy_attributes"),
                      ("trainer", "trainer_attributes")):
        missing += ["%s.%s" % (role, a) for a in orch[key] if not hasattr(roles[role], a)]
    for space, subs in orch["vec_env_space_attributes"].items():
        obj = getattr(learner.train_envs, space)[0]
        for sub in subs:
            if sub == "n":                                      # only read on a Discrete space; this env's is a Box
                continue
            if not hasattr(obj, sub):
                missing.append("vec_env.%s[0].%s" % (space, sub))
    if type(learner.train_envs.action_space[0]).__name__ != "Box":
        missing.append("vec_env.action_space[0] class name Box")
    for callee, shapes in orch["calls"].items():
        role, _, meth = callee.partition(".")
        if role in roles and "." not in meth:          # "<role>.<method>"; a dotted remainder is a module path (buffer.shared_buffer.X)
            fn = getattr(roles[role], meth, None)
        else:
            mod_name, _, fn_name = callee.rpartition(".")
            fn = getattr(importlib.import_module(mod_name), fn_name, None)
        if fn is None or not all(accepts(fn, s["positional"], s["keywords"]) for s in shapes):
            missing.append("call " + callee)

    # -- one reset + one step through the vec-env's numpy surface: shapes, dtypes and the per-env info keys ---------------------
    import numpy as np
    envs = learner.train_envs
    E, N = learner.rl_buffer.n_rollout_threads, cfg.num_agents
    obs = envs.reset()
    D = envs.observation_space[0].shape[0]
    ok = isinstance(obs, np.ndarray) and obs.shape == (E, N, D)
    obs, rew, dones, infos = envs.step(np.zeros((E, N, envs.action_space[0].shape[0]), np.float32))
    ok = ok and obs.shape == (E, N, D) and rew.shape == (E, N, 1) and dones.shape == (E, N) and dones.dtype == np.bool_ and len(infos) == E
    if not ok:
        missing.append("vec_env reset/step array contract")
    missing += ["info[%r]" % k for k in orch["info_keys"] if k not in infos[0]]
    if envs.share_observation_space[0].shape[0] != N * D:
        missing.append("share_observation_space = N * D")

    # -- run it --------------------------------------------------------------------------------------------------------------
    info = learner.rollout(learner.rl_buffer, learner.train_envs)
    missing += ["rollout()[%r]" % k for k in orch["rollout_info_keys"] if k not in info]
    for method in entry["learner_methods_called"]:
        getattr(learner, method)()
    print("B1_REPORT " + json.dumps({"missing": missing, "output_path": getattr(learner, "output_path", None)}))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
