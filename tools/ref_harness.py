"""Container-only harness that imports the *reference* (read-only, /root/reference) so that
golden vectors can be generated from the reference's own arithmetic.

Nothing in here travels as reference source: this file only (a) installs tiny in-memory stand-ins
for third-party *container classes* the reference imports but which carry no arithmetic
(`gym.Env`, `gym.spaces.Box` ...), and (b) subclasses the reference Scenario to override ONLY
`make_world`, because the shipped one hard-codes 4 agents / 20 landmarks
(uav_dcc_control/envs/mpe/multiagent/scenarios/coverage.py:40-41) and never forwards
comm_r_scale / comm_force_scale to CoverageWorld (coverage.py:30-34).  Every arithmetic method
(`CoverageWorld.step/update_connect/apply_connect_force/get_connect_force/integrate_state/
update_energy`, `Scenario.reset_world/reward/observation/done`, `MultiAgentEnv.step/_set_action`)
runs unmodified from /root/reference.

Used by tools/gen_golden.py only.  NOT importable on the GPU box (no /root/reference there).
"""
import os
import sys
import types

import numpy as np

REF_ROOT = os.environ.get("DCC_REFERENCE_ROOT", "/root/reference/uav_dcc_control")


def _install_gym_stub():
    if "gym" in sys.modules:
        return
    gym = types.ModuleType("gym")

    class Env(object):
        metadata = {}

        def close(self):
            pass

    class Space(object):
        pass

    class Box(Space):
        def __init__(self, low=None, high=None, shape=None, dtype=np.float32):
            if shape is None:
                shape = np.asarray(low).shape
            self.shape = tuple(shape)
            self.low = np.broadcast_to(np.asarray(low, dtype=dtype), self.shape)
            self.high = np.broadcast_to(np.asarray(high, dtype=dtype), self.shape)
            self.dtype = np.dtype(dtype)

    class Discrete(Space):
        def __init__(self, n):
            self.n = n
            self.shape = ()

    class Tuple(Space):
        def __init__(self, spaces):
            self.spaces = tuple(spaces)

    spaces = types.ModuleType("gym.spaces")
    spaces.Box, spaces.Discrete, spaces.Tuple, spaces.Space = Box, Discrete, Tuple, Space
    box = types.ModuleType("gym.spaces.box")
    box.Box = Box
    spaces.box = box
    envs = types.ModuleType("gym.envs")
    reg = types.ModuleType("gym.envs.registration")

    class EnvSpec(object):
        def __init__(self, *a, **k):
            pass

    reg.EnvSpec = EnvSpec
    reg.register = lambda *a, **k: None
    reg.load = lambda *a, **k: None
    envs.registration = reg
    gym.Env, gym.Space, gym.spaces, gym.envs = Env, Space, spaces, envs
    # picklable by reference (the reference's SubprocVecEnv sends the space objects through a Pipe, envs/wrappers.py:128)
    for cls, mod in ((Env, "gym"), (Space, "gym"), (Box, "gym.spaces"), (Discrete, "gym.spaces"), (Tuple, "gym.spaces"),
                     (EnvSpec, "gym.envs.registration")):
        cls.__module__, cls.__qualname__ = mod, cls.__name__
    sys.modules.update({"gym": gym, "gym.spaces": spaces, "gym.spaces.box": box,
                        "gym.envs": envs, "gym.envs.registration": reg})


def import_reference():
    """Put the reference on sys.path (front) and return the modules used by the fixtures."""
    _install_gym_stub()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import envs.mpe.multiagent.scenarios as scenarios  # noqa
    from envs.mpe.multiagent.CoverageWorld import CoverageWorld
    from envs.mpe.multiagent.core import Agent, Landmark
    from envs.mpe.multiagent.environment import MultiAgentEnv
    return scenarios, CoverageWorld, Agent, Landmark, MultiAgentEnv


def sized_scenario_class(extra_pois=None):
    """The reference Scenario with ONLY `make_world` replaced, constructed with the reference's own signature
    (num_agents, num_pois, r_cover, r_comm, comm_r_scale, comm_force_scale -- what DCEnv.__init__ passes, uav_dcc.py:21-29):
    make_world performs the same attribute assignments as coverage.py:46-59 for num_agents agents / num_pois landmarks, builds
    CoverageWorld(comm_r_scale, comm_force_scale) and sizes dist_mat NxN (CoverageWorld.py:11 hard-codes 4x4).
    `extra_pois` ([k,2] f64) is appended to pos_pois.npy when num_pois > 1000 (the file has 1000 rows)."""
    scenarios, CoverageWorld, Agent, Landmark, MultiAgentEnv = import_reference()
    ref_mod = scenarios.load("coverage.py")

    class SizedScenario(ref_mod.Scenario):
        def __init__(self, num_agents=4, num_pois=20, r_cover=0.25, r_comm=0.5, comm_r_scale=0.9, comm_force_scale=0.5):
            super().__init__(num_agents=num_agents, num_pois=min(num_pois, 1000), r_cover=r_cover, r_comm=r_comm,
                             comm_r_scale=comm_r_scale, comm_force_scale=comm_force_scale)
            if num_pois > 1000:
                assert extra_pois is not None and extra_pois.shape == (num_pois - 1000, 2)
                self.pos_pois = np.concatenate([self.pos_pois, extra_pois], 0)
            self.num_pois = num_pois

        def make_world(self):
            N, M = self.num_agents, self.num_pois
            world = CoverageWorld(comm_r_scale=self.comm_r_scale, comm_force_scale=self.comm_force_scale)
            world.collaborative = True
            world.agents = [Agent() for _ in range(N)]
            world.landmarks = [Landmark() for _ in range(M)]
            world.dist_mat = np.zeros([N, N])
            for i, agent in enumerate(world.agents):
                agent.name = "agent_%d" % i
                agent.collide = False
                agent.silent = True
                agent.size = self.size
                agent.r_cover = self.r_cover
                agent.r_comm = self.r_comm
                agent.max_speed = 0.5
            for i, landmark in enumerate(world.landmarks):
                landmark.name = "poi_%d" % i
                landmark.collide = False
                landmark.movable = False
                landmark.size = self.size
                landmark.m_energy = self.m_energy
            self.reset_world(world)
            return world

    return SizedScenario


def make_reference_env(N, M, r_cover, r_comm, comm_r_scale, comm_force_scale, extra_pois=None):
    """Build the reference MultiAgentEnv at arbitrary (N, M) around `sized_scenario_class` (every arithmetic method unmodified)."""
    scenarios, CoverageWorld, Agent, Landmark, MultiAgentEnv = import_reference()
    sc = sized_scenario_class(extra_pois)(N, M, r_cover, r_comm, comm_r_scale, comm_force_scale)
    world = sc.make_world()
    env = MultiAgentEnv(world=world, reset_callback=sc.reset_world, reward_callback=sc.reward,
                        observation_callback=sc.observation, done_callback=sc.done)
    return env, world, sc
