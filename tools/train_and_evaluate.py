"""Train the shipped task (4 UAV x 20 PoI) on the GPU for a number of iterations and report the metrics of the reference's
README curves (coverage rate, connectivity rate, steps needed to cover every PoI) with the deterministic policy.
usage: python tools/train_and_evaluate.py [iters] [envs] [curve.json]"""
import json, os, sys, time, yaml, torch
from argparse import Namespace
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); PKG = os.path.join(R, "dynamic-coverage-control_amd")
if len(sys.argv) > 3 and not os.path.isabs(sys.argv[3]):
    sys.argv[3] = os.path.join(R, sys.argv[3])
sys.path.insert(0, PKG); os.chdir(PKG)
import utils.pytorch_utils as ptu
ptu.set_gpu_mode(True, 0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150
envs = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
curve_path = sys.argv[3] if len(sys.argv) > 3 else None      # relative to the repo root
cfg = {}
for f in ("config/env_config/dcc.yaml", "config/algo_config/mappo.yaml", "config/expt.yaml"):
    cfg.update(yaml.safe_load(open(os.path.join(PKG, f))))
cfg.update(n_rollout_threads=envs, n_eval_rollout_threads=256, save_model=False, n_iters=iters, eval_interval=10 ** 9, log_interval=10 ** 9)
from learner import Learner
lr = Learner(Namespace(**cfg))
t0 = time.time()
curve = []
lr.warmup(lr.rl_buffer, lr.train_envs)
for it in range(1, iters + 1):
    lr.policy.lr_decay(it, iters)
    r = lr.rollout(lr.rl_buffer, lr.train_envs)
    lr.rl_update()
    if it % 25 == 0 or it == iters:
        curve.append({"iter": it, "s": round(time.time() - t0, 1), "reward": r["reward"], "coverage": r["coverage_rate"]})
        if it % 250 == 0 or it == iters:     # the README's curves: coverage rate and steps needed, sampled policy (like its test rollouts)
            ev = lr.evaluate(steps=150, deterministic=False)
            curve[-1].update(eval_coverage=ev["coverage_rate"], eval_solved=ev["solved_fraction"], eval_steps_to_cover=ev["steps_to_cover"])
            print("   evaluate(sampled) at iter %d: %s" % (it, {k: round(v, 4) for k, v in ev.items()}), flush=True)
        print("iter %d  %.1f s  rollout reward %.1f  coverage %.4f" % (it, time.time() - t0, r["reward"], r["coverage_rate"]), flush=True)
for det in (True, False):
    res = lr.evaluate(steps=150, deterministic=det)
    print("evaluate(deterministic=%s, 256 envs, 150 steps): %s" % (det, {k: round(v, 4) for k, v in res.items()}))

sweep = {}
for tau in (1.0, 0.75, 0.5, 0.25, 0.1, 0.03):
    sweep[str(tau)] = lr.evaluate(steps=150, noise_scale=tau)
    print("evaluate(noise_scale=%.2f): %s" % (tau, {k: round(v, 4) for k, v in sweep[str(tau)].items()}))
print("policy std (exp(logstd)): %s" % lr.policy.actor.act.action_out.logstd._bias.exp().tolist())
if curve_path:
    res = {"task": "shipped 4 UAV x 20 PoI (config/env_config/dcc.yaml)", "iters": iters, "envs": envs, "train_s": round(time.time() - t0, 1),
           "curve": curve, "final_sampled": lr.evaluate(steps=150, deterministic=False), "final_mean_action": lr.evaluate(steps=150, deterministic=True),
           "noise_scale_sweep": sweep, "policy_std": lr.policy.actor.act.action_out.logstd._bias.exp().tolist()}
    json.dump(res, open(curve_path, "w"), indent=1)
    lr.trainer.save_model(os.path.splitext(curve_path)[0] + "_model")
