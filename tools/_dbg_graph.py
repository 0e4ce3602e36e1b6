import sys, os, yaml, torch, numpy as np
from argparse import Namespace
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); PKG=os.path.join(R,'dynamic-coverage-control_amd'); sys.path.insert(0,PKG)
import utils.pytorch_utils as ptu
ptu.set_gpu_mode(True,0)
cfg={}
for f in ("config/env_config/dcc.yaml","config/algo_config/mappo.yaml","config/expt.yaml"): cfg.update(yaml.safe_load(open(os.path.join(PKG,f))))
cfg.update(num_agents=8,num_pois=64,n_rollout_threads=1024,n_eval_rollout_threads=0,max_ep_len=150,save_model=False,n_iters=1,use_hip_graph=(sys.argv[1]=="graph"),structured_input=(sys.argv[2]=="st"))
from learner import Learner
lr=Learner(Namespace(**cfg))
for i in range(5):
    info=lr.rollout(lr.rl_buffer,lr.train_envs)
    b=lr.rl_buffer
    print(i, info, "act std %.4f mean %.4f |a|>1 frac %.3f"%(float(b.actions.std()), float(b.actions.mean()), float((b.actions.abs()>1).float().mean())), "rew mean %.3f"%float(b.rewards.mean()), "masks0 %d"%int((b.masks==0).sum()), "act[0]==act[1]? %s"%bool(torch.equal(b.actions[0],b.actions[1])))
