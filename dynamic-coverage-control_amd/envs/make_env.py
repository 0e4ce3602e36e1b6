"""`make_env(cfg)`: the env factory Learner calls (reference: uav_dcc_control/envs/make_env.py:8-49).

The reference builds `cfg.n_rollout_threads` DCEnv objects and wraps them in Dummy/SubprocVecEnv;
here the same cfg keys build ONE batched HIP env with that many instances on the current GPU.  In
a multi-GPU job (one process per GPU) `n_rollout_threads` is the GLOBAL env count and each rank
builds its contiguous shard of n_rollout_threads / world_size envs.
"""
import utils.pytorch_utils as ptu
from envs.hip_vec_env import HipCoverageVecEnv


def make_env(cfg, **kwargs):
    for k, v in (kwargs or {}).items():
        setattr(cfg, k, v)
    if "uav_dcc" not in getattr(cfg, "env_file", "mpe.uav_dcc"):
        raise NotImplementedError("env_file: %s not found" % cfg.env_file)
    world = ptu.world_size()
    E = int(cfg.n_rollout_threads)
    rank = 0
    if world > 1:
        import torch.distributed as dist
        rank = dist.get_rank()
        if E % world:
            raise ValueError("n_rollout_threads (%d) must be divisible by the number of GPUs (%d)" % (E, world))
        E //= world
    dev = ptu.device.index if ptu.device.type == "cuda" else None
    return HipCoverageVecEnv(E, num_agents=cfg.num_agents, num_pois=cfg.num_pois, r_cover=cfg.r_cover,
                             r_comm=cfg.r_comm, comm_r_scale=cfg.comm_r_scale,
                             comm_force_scale=cfg.comm_force_scale, max_ep_len=cfg.max_ep_len, device=dev,
                             env0=rank * E, env_total=E * world)
