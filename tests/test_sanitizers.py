"""Sanitizer / race coverage.

CPU: the C restatement and its `_cpu` ABI twins (oracle/dcc_env_cpu.c) built with AddressSanitizer + UBSan and driven by
a small C program over odd sizes, both force branches, auto-resets, every optional output -- any out-of-bounds access,
leak or undefined behaviour fails the run.

GPU (-m gpu): the LDS hand-off of the multi-wave kernels (physics wave -> observation wave(s) through a double-buffered slot
guarded by ready/consumed counters) has no tool-based race detector on this stack, so it is soaked differentially: long
fused rollouts of the role-specialised and split kernels against the single-wave fused kernel, which has no hand-off, on
many envs and several action streams -- a lost or early hand-off shows up as a differing observation row or output."""
import os
import subprocess

import numpy as np
import pytest
import torch

from conftest import ROOT

DRIVER = r'''
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "dcc_env.h"
int dcc_env_create_cpu(const dcc_env_cfg*, dcc_env**); int dcc_env_destroy_cpu(dcc_env*); int dcc_env_obs_dim_cpu(const dcc_env*);
int dcc_env_reset_cpu(dcc_env*, float*, void*); int dcc_env_step_cpu(dcc_env*, const void*, int, const dcc_env_out*, void*);
int dcc_env_rollout_cpu(dcc_env*, int32_t, const float*, uint64_t, uint32_t, int32_t, int32_t, const dcc_env_out*, void*);
int dcc_env_get_state_cpu(dcc_env*, double*, double*, float*, uint8_t*, void*);
int dcc_env_set_state_cpu(dcc_env*, const double*, const double*, const float*, const uint8_t*, void*);
static int run(int E, int N, int M, double cfs, double r_comm, int K) {
    dcc_env_cfg c; memset(&c, 0, sizeof c);
    c.n_envs = E; c.n_agents = N; c.n_pois = M; c.device = -1;
    c.r_cover = 0.25; c.r_comm = r_comm; c.comm_r_scale = 0.95; c.comm_force_scale = cfs;
    c.dt = 0.1; c.damping = 0.25; c.max_speed = 0.5; c.sensitivity = 5.0; c.mass = 1.0; c.contact_margin = 1e-3; c.m_energy = 5.0;
    c.rew_cover = 75.0; c.rew_done = 1500.0; c.rew_out = -100.0; c.bound_soft = 1.0; c.bound_hard = 1.5;
    double* poi = malloc(sizeof(double) * 2 * M);
    for (int j = 0; j < 2 * M; j++) poi[j] = ((j * 2654435761u) % 2000) / 1000.0 - 1.0;
    c.poi_xy = poi;
    dcc_env* e = NULL;
    if (dcc_env_create_cpu(&c, &e) != 0) return 1;
    const int D = dcc_env_obs_dim_cpu(e);
    const size_t KE = (size_t)K * E;
    dcc_env_out o; memset(&o, 0, sizeof o);
    o.obs = malloc(sizeof(float) * KE * N * D); o.reward = malloc(4 * KE); o.done = malloc(KE); o.connect = malloc(KE);
    o.connect_s = malloc(KE); o.coverage = malloc(4 * KE); o.assign = malloc(KE * M); o.reward64 = malloc(8 * KE);
    o.state_pos = malloc(16 * KE * N); o.state_vel = malloc(16 * KE * N); o.state_energy = malloc(4 * KE * M); o.state_done = malloc(KE * M);
    float* obs0 = malloc(sizeof(float) * E * N * D);
    int rc = dcc_env_reset_cpu(e, obs0, NULL);
    rc |= dcc_env_rollout_cpu(e, K, NULL, 7, 0, 0, E, &o, NULL);                 /* generated actions, every output */
    float* act = malloc(sizeof(float) * (size_t)E * N * 2);
    for (int i = 0; i < E * N * 2; i++) act[i] = (i % 3) - 1.0f;                  /* constant pushes: runs envs out of bounds -> resets */
    dcc_env_out o1; memset(&o1, 0, sizeof o1); o1.reward = o.reward; o1.done = o.done;   /* sparse outputs */
    for (int k = 0; k < 40; k++) rc |= dcc_env_step_cpu(e, act, DCC_ACT_F32, &o1, NULL);
    double* pos = malloc(16 * (size_t)E * N); double* vel = malloc(16 * (size_t)E * N); float* en = malloc(4 * (size_t)E * M); uint8_t* dn = malloc((size_t)E * M);
    rc |= dcc_env_get_state_cpu(e, pos, vel, en, dn, NULL);
    rc |= dcc_env_set_state_cpu(e, pos, vel, en, dn, NULL);
    rc |= dcc_env_rollout_cpu(e, 3, NULL, 9, 5, 2, E + 2, NULL, NULL);            /* no outputs at all */
    int resets = 0; for (size_t i = 0; i < KE; i++) resets += o.done[i];
    printf("E=%d N=%d M=%d cfs=%.1f: rc=%d D=%d resets=%d r0=%.3f\n", E, N, M, cfs, rc, D, resets, o.reward64[0]);
    free(pos); free(vel); free(en); free(dn); free(act); free(obs0);
    free(o.obs); free(o.reward); free(o.done); free(o.connect); free(o.connect_s); free(o.coverage); free(o.assign); free(o.reward64);
    free(o.state_pos); free(o.state_vel); free(o.state_energy); free(o.state_done);
    dcc_env_destroy_cpu(e); free(poi);
    return rc;
}
int main(void) {
    int rc = 0;
    rc |= run(3, 5, 37, 0.5, 0.15, 60);      /* odd sizes, pull force on (both branches fire) */
    rc |= run(2, 1, 9, 0.0, 0.4, 20);        /* a single agent */
    rc |= run(1, 32, 1000, 0.5, 0.1, 8);     /* c5-like */
    rc |= run(4, 8, 64, 0.0, 0.4, 150);      /* c2 shape */
    return rc;
}
'''


def test_cpu_restatement_under_asan_and_ubsan(tmp_path):
    src = tmp_path / "drv.c"
    src.write_text(DRIVER)
    exe = str(tmp_path / "drv")
    cmd = ["gcc", "-std=c11", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-ffp-contract=off", "-I", os.path.join(ROOT, "include"), str(src), os.path.join(ROOT, "oracle", "dcc_env_cpu.c"),
           "-o", exe, "-lm"]
    b = subprocess.run(cmd, capture_output=True, text=True)
    assert b.returncode == 0, b.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "ERROR" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    assert r.stdout.count("rc=0") == 4
    import re
    assert sum(int(m) for m in re.findall(r"resets=(\d+)", r.stdout)) > 0        # the auto-reset path ran under the sanitizers


SOAK = [
    # N,  M,   E,    cfs, r_comm, K,   multi-wave kernel
    (8, 64, 2048, 0.0, 0.40, 400),     # roles <.., 8, 64> (compile-time sizes)
    (8, 64, 1500, 0.5, 0.20, 300),     # roles with the pull force
    (6, 50, 1001, 0.5, 0.25, 300),     # roles generic, odd env count (a workgroup with one env)
    (16, 256, 512, 0.0, 0.15, 200),    # split <4, .., 16, 256>
    (9, 130, 333, 0.5, 0.20, 200),     # split generic (3 PoI tiles, rows not a multiple of 4)
]


@pytest.mark.gpu
@pytest.mark.parametrize("N,M,E,cfs,r_comm,K", SOAK, ids=["roles-spec", "roles-force", "roles-generic-odd", "split-spec", "split-generic"])
def test_lds_handoff_kernels_soak_against_the_single_wave_kernel(N, M, E, cfs, r_comm, K, monkeypatch):
    import dcc_hip
    poi = np.random.RandomState(N + M).uniform(-1, 1, (M, 2))

    def make(single_wave):
        monkeypatch.setenv("DCC_NO_ROLES", "1" if single_wave else "0")
        monkeypatch.setenv("DCC_NO_SPLIT", "1" if single_wave else "0")
        return dcc_hip.HipCoverageEnv(E, N, M, poi, 0.25, r_comm, 0.95, cfs)

    multi, single = make(False), make(True)
    chunk = 50
    om = multi.alloc_out(chunk, reward64=True); om.update(multi.alloc_state_out(chunk))
    os_ = single.alloc_out(chunk, reward64=True); os_.update(single.alloc_state_out(chunk))
    for seed in (1, 2, 3):                                   # three action streams, state carried over between them
        for k0 in range(0, K, chunk):
            multi.rollout(chunk, seed=seed, step0=k0, out=om)
            single.rollout(chunk, seed=seed, step0=k0, out=os_)
            for name in om:
                assert torch.equal(om[name], os_[name]), (name, seed, k0)
    sm, ss = multi.get_state(), single.get_state()
    assert all(torch.equal(sm[k], ss[k]) for k in sm)
    assert int(os_["done"].sum()) >= 0
    multi.close(); single.close()
