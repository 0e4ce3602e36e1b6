#!/usr/bin/env python3
"""Container-only: extract the NAMES boundary B1-B3 consists of from the reference's entry point and
orchestrator and freeze them as tests/golden/b1_contract.json.

The reference's `train.py` and `learner.py` are the only callers of this package's drop-in surface.  What they need from it
is a set of identifiers: the modules they import, the functions / classes they take from them, the keyword names they call
them with, the YAML paths they open, the cfg attributes they read and assign, and the attributes / methods they touch on
the vec-env, the policy, the trainer and the rollout buffer.  This script walks the two files' syntax trees (`ast`; nothing
is imported or executed, no source text is kept) and writes those identifiers, sorted, to the JSON file; the tests
(tests/test_train_entry.py) then assert on the GPU box, where the reference does not exist, that the package provides every
one of them.  Names travel, text does not.

    python tools/check_b1_contract.py            # rewrite tests/golden/b1_contract.json
    python tools/check_b1_contract.py --check    # exit 1 if the committed file differs from a fresh extraction
"""
import argparse
import ast
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("DCC_REFERENCE", "/root/reference/uav_dcc_control")
OUT = os.path.join(ROOT, "tests", "golden", "b1_contract.json")

# receivers in learner.py, by role (the variable / attribute names the reference uses for each object)
ENV_RECEIVERS = {"self.train_envs", "self.test_envs", "self.render_envs", "r_envs"}
BUFFER_RECEIVERS = {"self.rl_buffer", "self.test_buffer", "self.render_buffer", "r_buffer"}
POLICY_RECEIVERS = {"self.policy", "self.trainer.policy"}
TRAINER_RECEIVERS = {"self.trainer"}
CFG_RECEIVERS = {"self.cfg", "cfg", "test_cfg", "render_cfg"}
FIRST_PARTY = ("utils", "learner", "buffer", "envs", "algos")


def dotted(node):
    """`a.b.c` for a Name / Attribute chain (subscripts are looked through: `x.space[0].shape` -> `x.space.shape`)."""
    parts = []
    while True:
        if isinstance(node, ast.Attribute):
            parts.append(node.attr)
            node = node.value
        elif isinstance(node, ast.Subscript):
            node = node.value
        elif isinstance(node, ast.Name):
            parts.append(node.id)
            return ".".join(reversed(parts))
        else:
            return None


def first_party_imports(tree):
    mods, aliases = {}, {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Import):
            for a in node.names:
                if a.name.split(".")[0] in FIRST_PARTY:
                    mods.setdefault(a.name, [])
                    aliases[a.asname or a.name] = a.name
        elif isinstance(node, ast.ImportFrom) and node.module and node.module.split(".")[0] in FIRST_PARTY:
            names = mods.setdefault(node.module, [])
            for a in node.names:
                names.append(a.name)
                aliases[a.asname or a.name] = node.module + "." + a.name
    return {k: sorted(set(v)) for k, v in mods.items()}, aliases


def attribute_uses(tree, receivers):
    """{attr: 'r' | 'w' | 'rw'} for every `<receiver>.<attr>` in the tree."""
    uses = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Attribute):
            base = dotted(node.value)
            if base in receivers:
                mode = "w" if isinstance(node.ctx, ast.Store) else "r"
                prev = uses.get(node.attr, "")
                uses[node.attr] = "".join(sorted(set(prev + mode)))
    return uses


def second_level(tree, receivers):
    """{attr: [sub-attributes]} -- e.g. observation_space -> [shape], action_space -> [n, shape, __class__]."""
    out = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Attribute) and isinstance(node.value, (ast.Attribute, ast.Subscript)):
            inner = node.value
            while isinstance(inner, ast.Subscript):
                inner = inner.value
            if isinstance(inner, ast.Attribute) and dotted(inner.value) in receivers:
                out.setdefault(inner.attr, set()).add(node.attr)
    return {k: sorted(v) for k, v in out.items()}


def calls(tree, aliases):
    """{callee: [{"positional": n, "keywords": [...]}, ...]} -- the distinct argument shapes of the call sites of first-party
    names / methods of the role receivers."""
    out = {}
    roles = [("vec_env", ENV_RECEIVERS), ("buffer", BUFFER_RECEIVERS), ("policy", POLICY_RECEIVERS), ("trainer", TRAINER_RECEIVERS)]
    for node in ast.walk(tree):
        if not isinstance(node, ast.Call):
            continue
        name = dotted(node.func)
        if name is None:
            continue
        key = None
        head, _, tail = name.partition(".")
        if name in aliases:
            key = aliases[name]
        elif head in aliases and tail:
            key = aliases[head] + "." + tail
        else:
            base, _, meth = name.rpartition(".")
            for role, recv in roles:
                if base in recv:
                    key = role + "." + meth
        if key is None:
            continue
        out.setdefault(key, set()).add((len(node.args), tuple(sorted(k.arg for k in node.keywords if k.arg))))
    return {k: [{"positional": n, "keywords": list(kw)} for n, kw in sorted(v)] for k, v in sorted(out.items())}


def extract():
    with open(os.path.join(REF, "train.py")) as f:
        t_tree = ast.parse(f.read())
    with open(os.path.join(REF, "learner.py")) as f:
        l_tree = ast.parse(f.read())

    t_mods, t_alias = first_party_imports(t_tree)
    yaml_paths = sorted({n.value for n in ast.walk(t_tree)
                         if isinstance(n, ast.Constant) and isinstance(n.value, str) and n.value.endswith(".yaml")})
    t_cfg = attribute_uses(t_tree, {"cfg"})
    learner_var_calls = sorted({n.func.attr for n in ast.walk(t_tree)
                                if isinstance(n, ast.Call) and isinstance(n.func, ast.Attribute) and dotted(n.func.value) == "learner"})
    # the order in which the YAML files are merged (later wins): positions of the loaded variables in the merge call
    loads, merge_order = {}, []
    for n in ast.walk(t_tree):
        if isinstance(n, ast.Assign) and isinstance(n.value, ast.Call) and dotted(n.value.func) == "OmegaConf.load":
            loads[n.targets[0].id] = n.value.args[0].value
        if isinstance(n, ast.Call) and dotted(n.func) == "OmegaConf.merge":
            merge_order = [a.id for a in n.args]
    merge_order = [loads[v] for v in merge_order]

    l_mods, l_alias = first_party_imports(l_tree)
    cls = next(n for n in l_tree.body if isinstance(n, ast.ClassDef) and n.name == "Learner")
    methods = {}
    for fn in cls.body:
        if isinstance(fn, ast.FunctionDef):
            methods[fn.name] = [a.arg for a in fn.args.args if a.arg != "self"]
    info_keys = sorted({k.value for n in ast.walk(l_tree) if isinstance(n, ast.Subscript) and dotted(n.value) == "info"
                        for k in [n.slice] if isinstance(k, ast.Constant)})
    rollout_keys = sorted({k.value for fn in cls.body if isinstance(fn, ast.FunctionDef) and fn.name == "rollout"
                           for r in ast.walk(fn) if isinstance(r, ast.Return) and isinstance(r.value, ast.Dict)
                           for k in r.value.keys if isinstance(k, ast.Constant)})
    return {
        "_about": "identifiers the reference's train.py / learner.py need from this package (tools/check_b1_contract.py; names only)",
        "train_py": {
            "imports": t_mods,
            "yaml_paths_in_merge_order": merge_order,
            "yaml_paths": yaml_paths,
            "cfg_attributes": t_cfg,
            "calls": calls(t_tree, t_alias),
            "learner_methods_called": learner_var_calls,
        },
        "learner_py": {
            "imports": l_mods,
            "learner_methods": methods,
            "cfg_attributes": attribute_uses(l_tree, CFG_RECEIVERS),
            "vec_env_attributes": attribute_uses(l_tree, ENV_RECEIVERS),
            "vec_env_space_attributes": second_level(l_tree, ENV_RECEIVERS),
            "buffer_attributes": attribute_uses(l_tree, BUFFER_RECEIVERS),
            "policy_attributes": attribute_uses(l_tree, POLICY_RECEIVERS),
            "trainer_attributes": attribute_uses(l_tree, TRAINER_RECEIVERS),
            "calls": calls(l_tree, l_alias),
            "info_keys": info_keys,
            "rollout_info_keys": rollout_keys,
        },
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    if not os.path.isdir(REF):
        sys.exit("reference not found at %s (this tool runs in the build container only)" % REF)
    text = json.dumps(extract(), indent=1, sort_keys=True) + "\n"
    if args.check:
        with open(OUT) as f:
            same = f.read() == text
        print("b1_contract.json is %s" % ("up to date" if same else "STALE"))
        sys.exit(0 if same else 1)
    with open(OUT, "w") as f:
        f.write(text)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
