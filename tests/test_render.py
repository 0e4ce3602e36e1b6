"""Headless stand-in for the reference's pyglet viewer (envs/render.py; reference environment.py:209-290, learner.py:195-210).
CPU: the rasteriser itself.  GPU: the vec-env's render("rgb_array") surface and the GIF a render rollout writes."""
import os
from argparse import Namespace

import numpy as np
import pytest

from conftest import PKG


def test_rasterize_draws_uavs_pois_and_links():
    from envs.render import rasterize, save_gif, POI_DONE, UAV, LINK
    poi = np.array([[0.5, 0.5], [-0.5, -0.5], [0.0, 0.9]])
    pos = np.array([[0.0, 0.0], [0.3, 0.0], [-1.0, 1.0]])
    f = rasterize(pos, poi, np.array([0.0, 5.0, 2.0]), np.array([0, 1, 0]), 0.2, 5.0, r_comm=0.4, size=200)
    assert f.shape == (200, 200, 3) and f.dtype == np.uint8
    has = lambda c: bool((f == np.array(c, np.uint8)).all(axis=2).any())
    assert has(UAV) and has(POI_DONE) and has(LINK)          # UAV 0 and 1 are within 2 r_comm of each other
    g = rasterize(pos, poi, np.zeros(3), np.zeros(3), 0.2, 5.0, r_comm=0.1, size=200)
    assert not bool((g == np.array(POI_DONE, np.uint8)).all(axis=2).any()) and not bool((g == np.array(LINK, np.uint8)).all(axis=2).any())
    # y is up: the UAV at (-1, +1) is in the upper-left quadrant of the image
    ys, xs = np.nonzero((f == np.array(UAV, np.uint8)).all(axis=2))
    assert ((ys < 100) & (xs < 100)).any()
    import tempfile
    from PIL import Image
    with tempfile.TemporaryDirectory() as d:
        save_gif([f, g, f], os.path.join(d, "a.gif"))
        assert Image.open(os.path.join(d, "a.gif")).n_frames == 3


@pytest.mark.gpu
def test_render_surface_and_gif_of_a_render_rollout(tmp_path):
    """B2: render(mode) exists like on the reference's vec-envs -- "human" is a no-op without a display, "rgb_array" returns
    frames indexed like the reference's (frame[0][0] = image of env 0) -- and with save_gifs the learner's render rollout
    (every render_interval iterations, n_render_rollout_threads envs) writes models_<iter>.gif where the reference does."""
    import yaml
    import utils.pytorch_utils as ptu
    ptu.set_gpu_mode(True, 0)
    from envs.hip_vec_env import HipCoverageVecEnv
    env = HipCoverageVecEnv(5, 4, 20)
    env.reset()
    assert env.render("human") is None
    fr = env.render("rgb_array")
    assert fr.shape == (5, 1, 350, 350, 3) and fr.dtype == np.uint8
    a = np.zeros((5, 4, 2), np.float32); a[:, 0] = (1.0, 0.0)
    for _ in range(10):
        env.step(a)
    fr2 = env.render("rgb_array")
    assert (fr2[0][0] != fr[0][0]).any()                                   # UAV 0 moved
    env.close()
    cfg = {}
    for f in ("config/env_config/dcc.yaml", "config/algo_config/mappo.yaml", "config/expt.yaml"):
        cfg.update(yaml.safe_load(open(os.path.join(PKG, f))))
    cfg.update(n_rollout_threads=8, n_eval_rollout_threads=0, n_render_rollout_threads=1, max_ep_len=7, n_iters=2, ppo_epoch=1,
               algo_hidden_size=32, save_model=True, save_interval=10, save_gifs=True, render_interval=2, main_save_path=str(tmp_path) + "/")
    from learner import Learner
    lr = Learner(Namespace(**cfg))
    lr.train()
    gif = os.path.join(lr.output_path, "models_2.gif")
    assert os.path.exists(gif)
    from PIL import Image
    assert Image.open(gif).n_frames == 7 and not os.path.exists(os.path.join(lr.output_path, "models_1.gif"))
    ptu.set_gpu_mode(False)
