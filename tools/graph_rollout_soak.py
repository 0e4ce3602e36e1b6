"""Soak test of the hipGraph-replayed rollout (use_hip_graph): two identical learners, one replaying the captured
rollout, one issuing it eagerly, same seeds -> parameters and rollout statistics must stay bit-identical over hundreds
of iterations (the replayed-graph reduction issue of tools/graph_reduce_probe.py shows up only after several hundred
replays).  usage: python tools/graph_rollout_soak.py [iters] [envs] [agents] [pois] [steps]"""
import os, sys, yaml, torch
from argparse import Namespace
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); PKG = os.path.join(R, "dynamic-coverage-control_amd")
sys.path.insert(0, PKG); os.chdir(PKG)
import utils.pytorch_utils as ptu
ptu.set_gpu_mode(True, 0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 400
envs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cfg = {}
for f in ("config/env_config/dcc.yaml", "config/algo_config/mappo.yaml", "config/expt.yaml"):
    cfg.update(yaml.safe_load(open(os.path.join(PKG, f))))
A = int(sys.argv[3]) if len(sys.argv) > 3 else 4
M = int(sys.argv[4]) if len(sys.argv) > 4 else 20
T = int(sys.argv[5]) if len(sys.argv) > 5 else 50
cfg.update(n_rollout_threads=envs, n_eval_rollout_threads=0, save_model=False, n_iters=iters, ppo_epoch=2 if envs < 1000 else 1,
           max_ep_len=T, num_agents=A, num_pois=M)
from learner import Learner
g = Learner(Namespace(**dict(cfg, use_hip_graph=True)))
e = Learner(Namespace(**dict(cfg, use_hip_graph=False)))
bad = 0
for it in range(1, iters + 1):
    out = []
    for lr in (g, e):
        lr.policy.lr_decay(it, iters)
        torch.manual_seed(10000 + it)
        r = lr.rollout(lr.rl_buffer, lr.train_envs)
        i = lr.rl_update()
        out.append((r, i))
    same = out[0] == out[1] and all(torch.equal(a, b) for a, b in zip(g.policy.actor.state_dict().values(), e.policy.actor.state_dict().values()))
    bad += (not same)
    if not same and bad < 4:
        print("MISMATCH at iteration", it, out[0][0], out[1][0])
    if it % 100 == 0:
        print("iteration %d: %d mismatching so far; reward %.1f coverage %.3f" % (it, bad, out[0][0]["reward"], out[0][0]["coverage_rate"]), flush=True)
print("soak: %d iterations x %d envs, graph-replayed vs eager rollouts: %d mismatching iterations" % (iters, envs, bad))
