#!/usr/bin/env python
"""bench.py -- agent-env-steps/s of the batched coverage-env hot path on MI355X.

Workload (BASELINE.json configs[1], "c2"): 8 UAVs x 64 PoIs x 4096 envs per GPU, random actions,
env-step HIP kernel only.  A "step" (--steps K / --warmup W) is ONE PASS of the hot path over one
batch of synthetic input: --launches-per-step (default 32) rollouts, each a fused launch of
--steps-per-launch (default T=150) batched env steps over all E envs that reads its actions
[T,E,N,2] from HBM and writes obs [T,E,N,D] + the per-step outputs to HBM, i.e. the full algorithmic
byte contract of SURVEY.md section 8d (11,851 B per env-step at c2).  One step is therefore
32 x 150 x 4096 x 8 = 157 M agent-env-steps and ~40 ms of GPU time, so the driver's
`--steps 20 --warmup 5` times a 0.8 s region of 640 launches.

  python bench.py --gpus N --steps K --warmup W
      (N>1 without WORLD_SIZE in the environment: re-executes itself under torch.distributed.run, one rank per GPU)

Prints ONE JSON line on rank 0.  `roofline` is always measured live: HIP events around every timed
launch on the launch stream.  `cpu_baseline` times the CPU oracle (oracle/dcc_oracle.c, "port") on
the host cores of the same box in the same run: rank 0, after the timed region, at every N (the other ranks of an N > 1 job wait
at a barrier; `ranks_waiting` says how many).  `c4` (unless --no-c3) is a bounded env-step leg at BASELINE configs[3]'s shape with the
job-wide env count fixed (16 UAV x 256 PoI x 8192 envs / N per GPU: strong scaling), `c5` the same for configs[4]
(32 UAV x 1024 PoI x 16384 envs / N, connectivity pull force on).  `c3` (unless --no-c3) is a bounded run of BASELINE configs[2]: full
MAPPO iterations (policy-driven rollout + HIP GAE + PPO epochs) at the same shape, with the RCCL
gradient all-reduce when N>1; it never affects `value`.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "dynamic-coverage-control_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable


def _host_threads():
    """Threads the host really gives us: affinity mask, capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per))))
    except Exception:
        pass
    return n


def _test_hook(name, default=None):
    """TEST-ONLY switches (DCC_BENCH_BACKEND, DCC_DIST_SINGLE, --test-kill-rank-at-leg): honoured only with DCC_TESTING=1 -- the one
    gate in front of every test seam (utils/pytorch_utils.test_hook); set without it they are an error, not a silent change."""
    v = os.environ.get(name)
    if v is None:
        return default
    if os.environ.get("DCC_TESTING") != "1":
        raise SystemExit("%s is a test hook: it is honoured only with DCC_TESTING=1 in the environment" % name)
    return v


def cpu_baseline(N, M, poi, r_cover, r_comm, crs, cfs, E=256, K=150, budget_all_s=8.0, budget_single_s=2.5, label="c2"):
    """the CPU restatement THROUGH THE SAME C-ABI as the GPU path (the `_cpu` twins of
    include/dcc_env.h, oracle/dcc_env_cpu.c: dcc_env_rollout_cpu with the product's dcc_env_cfg / dcc_env_out structs, host
    pointers) on a batch of E = 256 envs x K = 150 steps of this (N, M): once on a single thread, once on all host threads
    (one contiguous env range per thread).  In-kernel action stream, per-step reward / done / flags / coverage written,
    observation rows produced and cast to float32 like the GPU writes them.  Both legs are bounded: the all-thread leg repeats
    whole K-step passes until `budget_all_s` has passed (at least one), the single-thread leg stops after `budget_single_s`
    even if it has not finished its K steps (large shapes) -- `sample` says what was run."""
    from oracle import oracle
    oracle.build()
    cores = _host_threads()
    # steps per C call = the granularity of the time checks: about 0.1 s of single-thread work per call (the restatement costs
    # ~75 ns per UAV x PoI pair and env-step: c2 10 steps, c4 2, c5 1), so that the budgets below are real bounds at every shape
    pair_s = 75e-9 * N * M
    sub = int(max(1, min(10, round(0.1 / (E * pair_s)))))
    # single-thread leg at very large shapes: fewer envs, so that even one call stays near half a second
    E1 = int(max(8, min(E, 0.5 / (sub * pair_s))))

    def make(n_envs):
        e = oracle.CpuTwinEnv(n_envs, N, M, poi, r_cover, r_comm, crs, cfs)
        e.reset()
        return e, e.alloc_out(sub, obs=True, assign=True)

    # ---- single thread: E1 envs, K steps (or as many as fit the budget)
    o, out1 = make(E1)
    t0 = time.perf_counter()
    k1 = 0
    while k1 < K and (k1 == 0 or time.perf_counter() - t0 < budget_single_s):
        n = min(sub, K - k1)
        o.rollout(n, seed=0, step0=k1, env0=0, env_total=E1, out=out1)
        k1 += n
    dt1 = time.perf_counter() - t0
    rate1 = E1 * k1 * N / dt1
    o.close()
    del o, out1                                        # (c5: 1.7 GB of rows) before the per-thread buffers are built
    # ---- all threads: thread i owns the contiguous env range [i * E / cores, (i + 1) * E / cores); passes of K steps, the last one
    # cut at the deadline (at least one call per thread): what is counted is the steps actually done
    bounds = [E * i // cores for i in range(cores + 1)]
    envs = [make(bounds[i + 1] - bounds[i]) if bounds[i + 1] > bounds[i] else None for i in range(cores)]
    steps_done = [0] * cores
    deadline = [0.0]

    def work(i):
        if envs[i] is None:
            return
        e, out = envs[i]
        p = 0
        while True:
            for k in range(0, K, sub):                # ctypes releases the GIL inside the C call
                if steps_done[i] > 0 and time.perf_counter() >= deadline[0]:
                    return
                n = min(sub, K - k)
                e.rollout(n, seed=1 + p, step0=k, env0=bounds[i], env_total=E, out=out)
                steps_done[i] += n
            p += 1

    ths = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    t0 = time.perf_counter()
    deadline[0] = t0 + budget_all_s
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    env_steps = sum((bounds[i + 1] - bounds[i]) * steps_done[i] for i in range(cores))
    value = env_steps * N / dt
    for ev in envs:
        if ev is not None:
            ev[0].close()
    live = [sd for sd, ev in zip(steps_done, envs) if ev is not None]
    try:
        model = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    return {"value": value, "unit": "agent-env-steps/s", "cores": cores, "kind": "port",
            "sample": "oracle/dcc_env_cpu.c (the `_cpu` twins of include/dcc_env.h over the C restatement of the reference env, "
                      "float64): BASELINE.md 4.2 batch E = %d envs x K = %d steps of the %s workload (N=%d, M=%d%s), counter-based "
                      "random actions, observation rows written, %d step(s) per C call.  All %d host threads (one contiguous env range "
                      "each, %.1f s budget, the last pass cut at the deadline): %d-%d steps per thread (%.2f-%.2f passes of %d) in %.1f s = "
                      "%d env-steps.  Single thread (%.1f s budget): %d of the %d steps of %d envs in %.1f s = %.0f agent-env-steps/s.  "
                      "Host CPU: %s" % (E, K, label, N, M, ", pull force on" if cfs > 0 else "", sub, cores, budget_all_s,
                                        min(live), max(live), min(live) / K, max(live) / K, K, dt, env_steps, budget_single_s, k1, K, E1,
                                        dt1, rate1, model),
            "value_1core": rate1, "batch_envs": E, "batch_steps": K, "single_thread_steps_done": k1, "single_thread_envs": E1,
            "steps_per_call": sub, "all_thread_steps_done": [min(live), max(live)]}


def mappo_iterations(args, iters, warm_iters=2):
    """BASELINE config 3: N UAV x M PoI x E envs per GPU, T policy-driven env steps, HIP GAE scan and ppo_epoch
    full-batch PPO epochs per iteration (fp32, critic evaluated once per env).  Returns the measurement dict."""
    import yaml
    from argparse import Namespace
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import utils.pytorch_utils as ptu
    ptu.set_gpu_mode(True, torch.cuda.current_device())
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()        # peak_hbm_gb is this leg's own peak, not that of the legs before it
    cfg = {}
    for f in ("config/env_config/dcc.yaml", "config/algo_config/mappo.yaml", "config/expt.yaml"):
        cfg.update(yaml.safe_load(open(os.path.join(PKG, f))))
    cfg.update(num_agents=args.agents, num_pois=args.pois, n_rollout_threads=args.envs * world, n_eval_rollout_threads=0,
               max_ep_len=args.steps_per_launch, ppo_epoch=args.ppo_epoch, save_model=False, n_iters=1,
               comm_force_scale=args.comm_force_scale, r_comm=args.r_comm,
               use_hip_graph=not args.no_graph, compact_obs=not args.keep_rows, update_chunk_steps=args.update_chunk_steps,
               structured_input=not args.no_structured_input, num_mini_batch=args.num_mini_batch)
    from learner import Learner
    lr = Learner(Namespace(**cfg))

    def one_iter():
        t0 = time.perf_counter()
        info = lr.rollout(lr.rl_buffer, lr.train_envs)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        tinfo = lr.rl_update()
        torch.cuda.synchronize()
        return t1 - t0, time.perf_counter() - t1, info, tinfo

    for _ in range(warm_iters):  # hipBLASLt heuristics, allocator, eager rollout + hipGraph capture, first replay
        one_iter()
    dist = None
    if world > 1 or _test_hook("DCC_DIST_SINGLE") == "1":   # the latter: a 1-rank group over RCCL on a 1-GPU box (test hook)
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr = tu = 0.0
    for _ in range(iters):
        a, b, info, tinfo = one_iter()
        tr += a; tu += b
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt, tr, tu], dtype=torch.float64, device=ptu.device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, tr, tu = tt.tolist()
    E, N, T = args.envs, args.agents, args.steps_per_launch
    rows = T * E * N
    # algorithmic MLP FLOPs of one iteration: forward = 2 MACs per weight; the update is forward + two
    # backward GEMMs per layer.  "as_evaluated" = this build's formulation (critic once per env; first layers from the
    # compact features, 2M/N + 2N + 2 multiply-adds per output instead of D), "reference" = the reference's dense layers
    # with the critic row duplicated per agent.
    H, D = 256, 4 + 2 * (N - 1) + 5 * args.pois
    trunk = 2.0 * (H * H)
    ref_fwd = 2.0 * (D * H + 2 * H) + trunk + 2.0 * (N * D * H + H) + trunk
    structured = not args.no_structured_input
    l1a = 2.0 * ((2 * args.pois / N + 2 * N + 2) * H) if structured else 2.0 * D * H
    l1c = 2.0 * ((2 * args.pois + (2 * N + 4) * N) * H) if structured else 2.0 * N * D * H
    ours_fwd = l1a + trunk + 2.0 * 2 * H + (l1c + trunk + 2.0 * H) / N
    res = {"workload": "c3: %d UAV x %d PoI x %d envs per GPU, full MAPPO iteration = %d policy-driven env steps + HIP GAE "
                       "+ %d full-batch PPO epochs over %d agent rows; fp32 MLPs (hipBLASLt MFMA GEMMs + fused HIP tails), "
                       "f64 env state" % (N, args.pois, E, T, args.ppo_epoch, rows),
           "value": world * E * N * T * iters / dt, "unit": "agent-env-steps/s", "n_gpus": world, "iters_timed": iters,
           "iters_warmup": warm_iters, "s_per_iter": dt / iters, "rollout_s_per_iter": tr / iters,
           "update_s_per_iter": tu / iters, "rollout_agent_env_steps_per_sec": world * E * N * T / (tr / iters),
           "hip_graph_rollout": not args.no_graph, "rows_stored": bool(args.keep_rows), "structured_input": structured,
           "num_mini_batch": int(getattr(args, "num_mini_batch", 1)),
           "tuned_gemm_entries": lr.tuned_gemms,
           "tuned_gemm_warning": None if lr.tuned_gemms else (
               "0 entries of config/gemm_tunings_gfx950.csv are in use (its validators -- torch / ROCm / rocBLAS / hipBLASLt "
               "versions, GPU arch -- do not match this installation, or DCC_TUNED_GEMMS=0): the library heuristic picks the "
               "GEMM kernels, expect ~6 % less on this leg; regenerate with tools/tune_gemms.sh"),
           "grad_allreduce": ("%s x%d" % ({"nccl": "rccl"}.get(dist.get_backend(), dist.get_backend()), world)) if dist is not None else "none (1 GPU)",
           "mlp_tflop_per_iter_reference_formulation": ref_fwd * rows * (1 + 3 * args.ppo_epoch) / 1e12,
           "mlp_tflop_per_iter_as_evaluated": ours_fwd * rows * (1 + 3 * args.ppo_epoch) / 1e12,
           "peak_hbm_gb": torch.cuda.max_memory_allocated() / 1e9,
           "train_info": {k: float(v) for k, v in tinfo.items()}, "rollout_info": info}
    res["update_tflops_as_evaluated"] = ours_fwd * rows * 3 * args.ppo_epoch / 1e12 / (tu / iters)
    if (N, args.pois, E, T) == (8, 64, 4096, 150) and structured:      # north_star: MFMA utilisation of the policy GEMMs against the MI355X peak
        res["mfma"] = _mfma_file()
    lr.train_envs.close()
    del lr
    return res


def bench_mappo(args):
    """--mode mappo: config 3 as the whole line (not the headline)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    m = mappo_iterations(args, args.iters)
    T = args.steps_per_launch
    steps = args.iters * T
    res = {"metric": "agent_env_steps_per_sec", "value": m["value"], "unit": "agent-env-steps/s",
           "n_gpus": world, "steps": steps, "warmup": 2 * T, "ms_per_step": m["s_per_iter"] / T * 1e3, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f64 env state / f32 MLPs", "data": "synthetic",
           "config": m}
    if rank == 0:
        _emit_json(res)
    return res


def _agree(dist, backend, dev, ok):
    """True iff `ok` on every rank (one 1-element MIN all-reduce; no-op single process).  Called before a leg enters its
    first collective, so a rank whose local set-up failed makes ALL ranks skip the leg instead of leaving the others
    waiting in a barrier until the watchdog fires."""
    if dist is None:
        return bool(ok)
    t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float32, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item() > 0.5)


def _traffic_file(key, match):
    """Offline rocprofv3 PMC traffic of a workload (profiles/rNN/traffic_<key>.json, newest round first), or (None, None)."""
    import glob
    rounds = sorted((os.path.basename(os.path.dirname(f)) for f in glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*", "traffic_%s.json" % key))),
                    reverse=True)          # whatever rounds hold one, newest first (r06 > r05 > ...)
    for rnd in rounds:
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", rnd, "traffic_%s.json" % key)))
            if match(tj["workload"]):
                return tj, "offline rocprofv3 --pmc passes of the same launch (profiles/%s/traffic_%s.json), not measured in this run" % (rnd, key)
        except Exception:
            pass
    return None, None


def _mfma_file():
    """MFMA-pipe utilisation of the c3 iteration from the newest OFFLINE counter pass (profiles/rNN/mappo_c3_mfma_pmc.txt, written by
    tools/pmc_mfma_c3.sh: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE, its own run), or None."""
    import glob
    import re
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*", "mappo_c3_mfma_pmc.txt")), reverse=True):
        try:
            gemms, summary = [], None
            for line in open(f):
                m = re.match(r"^(Cijk_\S+)\s+(\d+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+\(([0-9.]+)% of GPU-active", line)
                if m:
                    gemms.append({"kernel": m.group(1)[:48], "calls": int(m.group(2)), "mfma_busy": float(m.group(3)), "tflops": float(m.group(5)),
                                  "ms_per_call": float(m.group(6)), "share_of_gpu_active": float(m.group(7)) / 100.0})
                m = re.match(r"^all library GEMMs.*busy ([0-9.]+) of their GPU-active time; whole iteration: ([0-9.]+) \(GEMMs are ([0-9.]+)%", line)
                if m:
                    summary = [float(v) for v in m.groups()]
            if gemms and summary:
                rnd = os.path.basename(os.path.dirname(f))
                return {"gemm_mfma_busy": summary[0], "iteration_mfma_busy": summary[1], "gemm_share_of_gpu_active": summary[2] / 100.0,
                        "largest_gemms": gemms[:4], "peak_tflops_fp32_matrix": 157.3,
                        "source": "offline rocprofv3 --pmc pass of `bench.py --mode mappo --iters 1 --ppo-epoch 2` (profiles/%s/mappo_c3_mfma_pmc.txt, "
                                  "tools/pmc_mfma_c3.sh), not measured in this run; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), "
                                  "tflops = SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 / kernel time of the trace run next to it" % rnd}
        except Exception:
            pass
    return None


def env_shape_leg(N, M, E_total, world, rank, local_dev, dist, backend, T=30, launches=160, warm=3, cfs=0.0, r_comm=0.4,
                  name="c4 (BASELINE configs[3])", cpu=None, place_tries=4, key="c4"):
    """Bounded env-step leg at another BASELINE shape with the JOB-WIDE env count fixed (strong scaling): BASELINE
    configs[3] is 16 UAV x 256 PoI x 8192 envs over the GPUs of the job, i.e. 8192 / world envs per GPU, no data-path
    collective.  `launches` fused launches of T steps (in-kernel action stream), HIP-event timed; value = job-wide
    agent-env-steps / max-over-ranks wall time."""
    import dcc_hip
    if E_total % world:
        raise ValueError("%d envs do not divide over %d GPUs" % (E_total, world))
    E = E_total // world
    from envs.hip_vec_env import load_pois       # the reference's PoI table (+ seeded synthetic rows beyond its 1000)
    dev = torch.device("cuda", local_dev)
    env = out = err = placement = None
    try:                                         # local set-up (allocations): no collective in here
        env = dcc_hip.HipCoverageEnv(E, N, M, load_pois(M), 0.2, r_comm, 0.95, cfs, device=local_dev)
        out = env.alloc_out(T, placed=place_tries)
        placement = env.placement_info
        env.reset()
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001
        err = e
    if not _agree(dist, backend, dev, err is None):      # every rank skips the leg together
        if env is not None:
            env.close()
        del env, out
        torch.cuda.empty_cache()
        raise RuntimeError("set-up failed on %s: %s" % ("this rank" if err is not None else "another rank", err))

    def go(n, step0, events=None):
        for i in range(n):
            if events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            env.rollout(T, seed=0, step0=step0 + i * T, env0=rank * E, env_total=E_total, out=out)
            if events is not None:
                e1.record(); events.append((e0, e1))

    go(warm, 0)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev = []
    t0 = time.perf_counter()
    go(launches, warm * T, ev)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms = [a.elapsed_time(b) for a, b in ev]
    bstep = dcc_hip.bytes_per_step(N, M, with_actions=False, with_obs=True)
    ach = bstep * E * T / (sum(ms) / len(ms) * 1e-3) / 1e9
    # HBM bytes the counters saw per env-step of this shape (offline passes), scaled to this launch
    tj, tsrc = _traffic_file(key, lambda w: (w["n_agents"], w["n_pois"]) == (N, M) and w.get("actions") == "rng")
    traffic = tj["traffic_bytes_per_env_step"] * E * T if tj else None
    env.close()
    del env, out
    torch.cuda.empty_cache()
    extra = {}
    if cpu is not None and rank == 0:
        # the CPU restatement at THIS (N, M) through the same C-ABI twins, single thread and all threads; in an
        # N > 1 job the other ranks wait at the barrier below (idle GPUs, the host threads are rank 0's)
        try:
            extra["cpu_baseline"] = cpu_baseline(N, M, load_pois(M), 0.2, r_comm, 0.95, cfs, label=name, **cpu)
            extra["cpu_baseline"]["ranks_waiting"] = world - 1
        except Exception as e:  # noqa: BLE001
            extra["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    if cpu is not None and dist is not None:
        dist.barrier()
    return {**extra,
            "workload": "%s: %d UAV x %d PoI x %d envs job-wide = %d per GPU over %d GPU(s), random-action env-step kernel%s, "
                        "%d fused launches x %d steps, actions drawn in-kernel, obs written"
                        % (name, N, M, E_total, E, world, ", connectivity pull force on (comm_force_scale %.1f, r_comm %.2f)" % (cfs, r_comm) if cfs > 0 else "",
                           launches, T),
            "value": E_total * N * T * launches / dt, "unit": "agent-env-steps/s", "scaling": "strong", "n_gpus": world,
            "envs_per_gpu": E, "us_per_step": sum(ms) / len(ms) / T * 1e3,
            "timed_region_s": dt,
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": tsrc,
                         "frac_physical": (traffic / (sum(ms) / len(ms) * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "bytes_per_env_step": bstep, "launch_ms_avg": sum(ms) / len(ms), "launch_ms_min": min(ms),
                         "launch_ms_max": max(ms), "launches_timed": len(ms), "output_placement": placement}}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _self_launch(n):
    """`python bench.py --gpus N` without a launcher: become `python -m torch.distributed.run ... bench.py <same args>`,
    one rank per GPU (rendezvous on 127.0.0.1)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush(); sys.stderr.flush()
    os.execv(sys.executable, cmd)


_JSON_OUT = None
_EMIT_LOCK = threading.Lock()
_EMITTED = False


def _claim_stdout():
    """stdout carries the ONE JSON line and nothing else: file descriptor 1 is pointed at stderr for everything this process
    (Python or native libraries) prints on the way, the line itself goes to the original descriptor."""
    global _JSON_OUT
    if _JSON_OUT is None:
        sys.stdout.flush()
        _JSON_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
        sys.stdout = sys.stderr


def _emit_json(res):
    """The ONE JSON line: whoever comes first (the main thread or the watchdog) writes it, anyone later is a no-op."""
    global _EMITTED
    with _EMIT_LOCK:
        if _EMITTED:
            return False
        line = None
        for _ in range(5):      # a helper thread (watchdog / SIGTERM) may serialise while the main thread is adding a leg's result
            try:
                line = json.dumps(dict(res))
                break
            except RuntimeError:        # "dictionary changed size during iteration"
                time.sleep(0.01)
        if line is None:
            line = json.dumps({k: res[k] for k in list(res) if not isinstance(res[k], dict)})      # the headline scalars at least
        out = _JSON_OUT or sys.stdout
        out.write(line + "\n")
        out.flush()
        _EMITTED = True
        return True


def rccl_proof(dist, backend, dev, rank, world, local_rank):
    """What the N>1 line says about the job it ran as (rank 0 returns the dict): every rank contributes its identity
    (all_gather_object), and an all-reduce of ones over the data-path backend must come back as the rank count."""
    p = torch.cuda.get_device_properties(dev)
    me = {"rank": rank, "local_rank": local_rank, "device": int(dev.index), "name": p.name,
          "uuid": str(getattr(p, "uuid", "")), "pci_bus_id": getattr(p, "pci_bus_id", None), "pid": os.getpid()}
    seen = [None] * world
    dist.all_gather_object(seen, me)
    ones = torch.ones(1024, dtype=torch.float32, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(ones)
    try:
        ver = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:  # noqa: BLE001
        ver = None
    uu = [r["uuid"] or "%s:%s" % (r["pci_bus_id"], r["device"]) for r in seen]
    # (the gloo TEST hook stages device tensors through the host, blocking: its collectives neither overlap compute nor compare with RCCL's)
    via_host = dist.get_backend() == "gloo" and os.environ.get("DCC_GLOO_VIA_HOST", "1") != "0"
    return {"backend": {"nccl": "rccl"}.get(dist.get_backend(), dist.get_backend()), "rccl_version": ver, "gloo_via_host": via_host,
            "world_size": dist.get_world_size(), "ranks_seen": seen, "distinct_devices": len(set(uu)),
            "allreduce_of_ones": float(ones[0].item()), "allreduce_ok": bool((ones == float(world)).all().item())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed steps; one step = --launches-per-step fused launches")
    ap.add_argument("--warmup", type=int, default=5, help="untimed warm-up steps")
    ap.add_argument("--launches-per-step", type=int, default=32,
                    help="rollouts (fused launches of --steps-per-launch env steps) in one bench step")
    ap.add_argument("--steps-per-launch", type=int, default=150)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--agents", type=int, default=8)
    ap.add_argument("--pois", type=int, default=64)
    ap.add_argument("--comm-force-scale", type=float, default=0.0)
    ap.add_argument("--r-comm", type=float, default=0.4)
    ap.add_argument("--actions", choices=["hbm", "rng"], default="hbm",
                    help="hbm: read pre-generated actions [T,E,N,2] (full byte contract); rng: draw in-kernel")
    ap.add_argument("--place-tries", type=int, default=12,
                    help="candidate allocations of the observation buffer to time before the run (0 = take the first)")
    ap.add_argument("--first-alloc-launches", type=int, default=48,
                    help="launches timed into the first-allocated buffer for roofline.first_allocation (outside the timed region)")
    ap.add_argument("--no-obs", action="store_true", help="skip the obs write (state-only variant, not the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--comm-r-scale", type=float, default=0.95, help="0 disables the connectivity flags (profiling aid)")
    ap.add_argument("--no-assign", action="store_true", help="skip the PoI-assignment output (profiling aid)")
    ap.add_argument("--no-scalars", action="store_true", help="skip reward/done/connect/coverage outputs (profiling aid)")
    ap.add_argument("--mode", choices=["env", "mappo"], default="env",
                    help="env: BASELINE config 2 (headline) + a bounded c3 leg; mappo: config 3 only as the whole line")
    ap.add_argument("--no-c3", action="store_true", help="--mode env: skip the bounded config-3 (MAPPO) leg")
    ap.add_argument("--c3-iters", type=int, default=2, help="--mode env: timed MAPPO iterations of the c3 leg")
    ap.add_argument("--leg-place-tries", type=int, default=4, help="candidate output allocations timed per bounded leg (c2_strong / c4 / c5)")
    ap.add_argument("--c3-timeout", type=float, default=240.0, help="give up on the c3 leg after this many seconds")
    ap.add_argument("--test-kill-rank-at-leg", default="", help=argparse.SUPPRESS)   # tests/test_bench_contract.py only: "<rank>:<leg>"
    ap.add_argument("--iters", type=int, default=2, help="--mode mappo: timed training iterations")
    ap.add_argument("--ppo-epoch", type=int, default=15)
    ap.add_argument("--num-mini-batch", type=int, default=1, help="mappo: the reference's num_mini_batch (1 = the shipped full-batch update)")
    ap.add_argument("--no-graph", action="store_true", help="mappo: issue the rollout eagerly from Python")
    ap.add_argument("--keep-rows", action="store_true",
                    help="mappo: also store the observation rows in the rollout buffer (default: compact env state only)")
    ap.add_argument("--no-structured-input", action="store_true",
                    help="mappo: dense first layers on the observation rows (reference formulation)")
    ap.add_argument("--update-chunk-steps", type=int, default=0,
                    help="mappo: >0 = chunked full-batch PPO step with gradient accumulation")
    args = ap.parse_args()
    if args.no_structured_input:
        args.keep_rows = True

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        _self_launch(args.gpus)
    _claim_stdout()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus (%d) != WORLD_SIZE (%d)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the env hot path has no CPU fallback")
    # DCC_BENCH_BACKEND=gloo is a test hook: several ranks may then share one GPU (rendezvous over gloo)
    backend = _test_hook("DCC_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        os.environ.setdefault("DCC_DIST_BACKEND", "gloo")
    local_dev = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    dist = None
    if world > 1 or _test_hook("DCC_DIST_SINGLE") == "1":   # the latter: a 1-rank group over RCCL on a 1-GPU box (test hook)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if args.mode == "mappo":
        import utils.pytorch_utils as ptu
        ptu.set_gpu_mode(True, local_dev)
        res = bench_mappo(args)
        if dist is not None:
            dist.destroy_process_group()
        return res

    import dcc_hip

    E, N, M, T, L = args.envs, args.agents, args.pois, args.steps_per_launch, args.launches_per_step
    r_cover, crs, cfs, r_comm = 0.2, args.comm_r_scale, args.comm_force_scale, args.r_comm
    poi_all = np.load(os.path.join(PKG, "envs", "mpe", "pos_pois.npy"))
    if M > len(poi_all):
        poi_all = np.concatenate([poi_all, np.random.RandomState(2024).uniform(-1, 1, (M - len(poi_all), 2))])
    poi = poi_all[:M]
    env = dcc_hip.HipCoverageEnv(E, N, M, poi, r_cover, r_comm, crs, cfs, device=local_dev)
    kernel_choice = env.kernel_choice()      # roles vs fused, measured by dcc_env_create on this box
    # the observation buffer is the best-placed of up to --place-tries candidate allocations (the same launch streams 6-8 % slower
    # into some allocations than into others of the same process: HipCoverageEnv.alloc_placed_obs, tools/placement_probe.py)
    # ... and next to it the figure for the FIRST allocation the process gets, which is what a caller that does not probe streams
    # into (the learner's default): allocated first, timed over --first-alloc-launches launches outside the timed region
    out_first = env.alloc_out(T, obs=not args.no_obs, assign=not args.no_assign) if (args.place_tries > 0 and not args.no_obs) else None
    # (the first allocation is candidate 0 of the probe: `value` never comes from a worse-placed buffer than `first_allocation`)
    out = env.alloc_out(T, obs=not args.no_obs, assign=not args.no_assign, placed=args.place_tries,
                        first_obs=out_first["obs"] if out_first is not None else None)
    placement = env.placement_info
    env.reset()
    if args.no_scalars:
        out = {k: v for k, v in out.items() if k in ("obs", "assign")}
    actions = None
    if args.actions == "hbm":
        # synthetic action stream, a different one per rank (the oracle is only the cpu_baseline leg's business)
        acts = np.random.default_rng(1000 + rank).uniform(-1.0, 1.0, (T, E, N, 2)).astype(np.float32)
        actions = torch.from_numpy(acts).to(dev)

    def run(n_steps, events=None, into=None, n_launches=None):
        """n_steps bench steps = n_steps * L fused launches of T env steps, back to back on the current stream."""
        step0 = 0
        for _ in range(n_steps * L if n_launches is None else n_launches):
            if events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            env.rollout(T, actions=actions, seed=0, step0=step0, env0=rank * E, env_total=world * E, out=out if into is None else into)
            if events is not None:
                e1.record()
                events.append((e0, e1))
            step0 += T

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    run(args.warmup)
    first_alloc = None
    if out_first is not None:
        if args.no_scalars:
            out_first = {k: v for k, v in out_first.items() if k in ("obs", "assign")}
        run(0, into=out_first, n_launches=4)
        torch.cuda.synchronize()
        fe = []
        run(0, events=fe, into=out_first, n_launches=args.first_alloc_launches)
        torch.cuda.synchronize()
        fms = [a.elapsed_time(b) for a, b in fe]
        first_alloc = sum(fms) / len(fms)
        del out_first
        torch.cuda.empty_cache()
        env.reset()
        run(1)                                   # back in the steady state of the main buffer
    barrier()
    events = []
    t0 = time.perf_counter()
    run(args.steps, events)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    env_steps = args.steps * L * T          # batched env steps in the timed region
    shape_name = {(8, 64, 4096): "c2 (BASELINE configs[1])", (16, 256, 1024): "c4 per-GPU shard (BASELINE configs[3]: 8192 envs over 8 GPUs)",
                  (32, 1024, 2048): "c5 per-GPU shard (BASELINE configs[4]: 16384 envs over 8 GPUs%s)" % ("" if cfs > 0 else ", pull force OFF"),
                  (4, 16, 4096): "c1 size (BASELINE configs[0] shape) batched"}.get((N, M, E), "custom shape")
    value = world * E * N * env_steps / dt
    res = {
        "metric": "agent_env_steps_per_sec", "value": value, "unit": "agent-env-steps/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s: %d UAV x %d PoI x %d envs per GPU, random-action env-step HIP kernel "
                               "only; `value` is WEAK scaling: %d envs on EVERY GPU (%d job-wide) -- the STRONG-scaling figure with BASELINE's 4096 "
                               "envs fixed job-wide is this line's `c2_strong` object (N > 1 only); one bench step = %d rollouts = %d fused launches "
                               "x %d batched env steps; actions %s, obs %s" % (
                                   shape_name, N, M, E, E, world * E, L, L, T,
                                   "read from HBM [T,E,N,2] f32" if actions is not None else "drawn in-kernel",
                                   "skipped" if args.no_obs else "written to HBM [T,E,N,D] f32"),
                   "n_agents": N, "n_pois": M, "envs_per_gpu": E, "global_envs": world * E,
                   "steps_per_launch": T, "launches_per_step": L, "env_steps_per_step": L * T,
                   "agent_env_steps_per_step": world * E * N * L * T, "timed_region_s": dt,
                   "comm_force_scale": cfs, "parallelism": "env-shard x%d (no data-path collective)" % world},
    }
    ms = [a.elapsed_time(b) for a, b in events]
    avg_ms = sum(ms) / len(ms)
    bstep = dcc_hip.bytes_per_step(N, M, with_actions=actions is not None, with_obs=not args.no_obs)
    alg = bstep * E * T
    ach = alg / (avg_ms * 1e-3) / 1e9
    # HBM bytes per launch: OFFLINE rocprofv3 PMC passes of this workload, not this run
    tj, traffic_src = _traffic_file("c2", lambda w: (w["n_agents"], w["n_pois"], w["envs"], w["steps_per_launch"]) == (N, M, E, T) and
                                    (w["actions"] == "hbm") == (actions is not None) and not args.no_obs)
    traffic = tj["traffic_bytes_per_launch"] if tj else None
    sms = sorted(ms)
    res["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                       "algorithmic_bytes_per_launch": alg,
                       "kernel": ("dcc_env_kernel<1,ACT,FORCE,NC,MC> (fused: one wave per env)" if (kernel_choice["choice"] == "fused" or os.environ.get("DCC_NO_ROLES") == "1")
                                  else "dcc_env_roles_kernel<ACT,FORCE,NC,MC> (a physics + an observation wave per two envs)") + " -- c2: <..,0,false,8,64>",
                       "kernel_choice": kernel_choice,
                       "output_placement": placement,      # untimed preparation, like the inputs: which allocation the rows go to
                       # `value` / `frac` above: the placement-probed buffer when --place-tries > 0.  The same launch into the
                       # process's FIRST allocation (what a caller that does not probe gets), timed outside the timed region:
                       "value_buffer": "placement-probed (best of the first allocation + up to %d further candidates)" % args.place_tries if args.place_tries > 0 else "first allocation",
                       "first_allocation": ({"launch_ms_avg": first_alloc, "launches_timed": args.first_alloc_launches,
                                             "achieved": alg / (first_alloc * 1e-3) / 1e9, "frac": alg / (first_alloc * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                             "value_equivalent": E * N * T / (first_alloc * 1e-3) * world}
                                            if first_alloc else None),
                       "bytes_per_env_step": bstep, "timing": "HIP events around every timed launch on the launch stream",
                       "launch_ms_avg": avg_ms, "launch_ms_min": sms[0], "launch_ms_median": sms[len(sms) // 2],
                       "launch_ms_max": sms[-1], "launches_timed": len(ms), "frac_of_achievable_6300": ach / 6300.0,
                       # the PHYSICAL rate: HBM bytes the counters saw per launch / launch time.  `frac` counts SURVEY 8d's
                       # algorithmic bytes, which include per-step state reads / writes a fused K-step launch keeps in
                       # registers, so it sits ~8 % above the physical fraction at c2
                       "physical_gbs": (traffic / (avg_ms * 1e-3) / 1e9) if traffic else None,
                       "frac_physical": (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None}
    res["config"]["steps_unit"] = ("--steps / --warmup / `steps` / `warmup` count PASSES over one synthetic batch (= %d fused launches x %d "
                                   "batched env steps each), not single env steps" % (L, T))
    if dist is not None:
        proof = rccl_proof(dist, backend, dev, rank, world, local_rank)
        res["rccl"] = proof
    env.close()
    del env, out, actions
    torch.cuda.empty_cache()
    if not args.no_cpu_baseline:
        # north_star: the reference-semantics CPU env timed on the host cores of the same box IN THE SAME RUN, at every N.  Rank 0
        # times the `_cpu` twins after the timed region; in an N > 1 job the other ranks wait at the barrier (their GPUs idle,
        # the host threads are rank 0's) and the budget is tighter (<= 10 s against ~10.5 s at N = 1).
        if rank == 0:
            try:
                res["cpu_baseline"] = (cpu_baseline(N, M, poi, r_cover, r_comm, crs, cfs) if world == 1 else
                                       cpu_baseline(N, M, poi, r_cover, r_comm, crs, cfs, budget_all_s=6.0, budget_single_s=2.5))
                res["cpu_baseline"]["ranks_waiting"] = world - 1
            except Exception as e:  # noqa: BLE001  (the headline must survive a failing baseline)
                res["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        if dist is not None:
            dist.barrier()

    def emit():
        if rank == 0:
            _emit_json(res)

    if not args.no_c3:
        # bounded legs at the other BASELINE configs.  They must never cost the headline line: a watchdog prints the line
        # with what is there and ends the process if a leg hangs (e.g. a collective that never completes), exceptions are
        # recorded as text, and a leg whose local set-up fails on one rank is skipped by all ranks (_agree).
        done = threading.Event()
        current = ["-"]

        def watchdog():
            if not done.wait(args.c3_timeout):
                res.setdefault(current[0], {"error": "timed out after %.0f s (legs share one budget)" % args.c3_timeout})
                emit()
                os._exit(0)

        threading.Thread(target=watchdog, daemon=True).start()

        # A rank that dies inside a leg makes the launcher SIGTERM the others; rank 0 may then sit in a collective that will never
        # complete, where a Python-level signal handler does not run.  The wake-up descriptor is written by the C-level handler
        # whatever the main thread is doing; a helper thread reads it, prints the line with what is there and ends the process.
        import signal
        sig_r, sig_w = os.pipe()
        os.set_blocking(sig_w, False)
        signal.signal(signal.SIGTERM, lambda *a: None)
        signal.set_wakeup_fd(sig_w, warn_on_full_buffer=False)

        def on_sigterm():
            while True:
                b = os.read(sig_r, 1)
                if b and b[0] == signal.SIGTERM:
                    res.setdefault(current[0], {"error": "terminated by the launcher (SIGTERM) during this leg: another rank failed"})
                    emit()
                    os._exit(128 + signal.SIGTERM)      # the line is out, the job still failed

        threading.Thread(target=on_sigterm, daemon=True).start()
        kill = args.test_kill_rank_at_leg or ""            # TEST-ONLY flag "<rank>:<leg>": that rank exits at the start of that leg
        if kill and os.environ.get("DCC_TESTING") != "1":
            raise SystemExit("--test-kill-rank-at-leg is a test hook: it is honoured only with DCC_TESTING=1 in the environment")

        def leg(key, fn):
            current[0] = key
            if kill == "%d:%s" % (rank, key):
                os._exit(17)
            try:
                res[key] = fn()
            except Exception as e:  # noqa: BLE001
                res[key] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
                torch.cuda.empty_cache()

        if world > 1:   # BASELINE's metric reads "4096 envs; 1/2/4/8 GPU": the fixed-4096 (STRONG) c2 figure next to the weak `value`
            leg("c2_strong", lambda: env_shape_leg(N, M, 4096, world, rank, local_dev, dist, backend, T=T, launches=16, warm=4, cfs=cfs,
                                                   r_comm=r_comm, name="c2 strong (BASELINE configs[1] with the job-wide env count fixed)", key="c2_strong",
                                                   place_tries=args.leg_place_tries))
        # >= 0.5 s timed on one GPU: 160 launches x 30 steps at ~117 us per step (c4), 72 x 4 at ~1.9 ms per step (c5)
        leg("c4", lambda: env_shape_leg(16, 256, 8192, world, rank, local_dev, dist, backend, launches=160, place_tries=args.leg_place_tries,
                                        cpu=None if args.no_cpu_baseline else dict(budget_all_s=3.0, budget_single_s=1.5)))
        # BASELINE configs[4]: the branchy wavefront path (pull force on), 16384 envs job-wide; 664 KB of rows per env-step
        leg("c5", lambda: env_shape_leg(32, 1024, 16384, world, rank, local_dev, dist, backend, T=4, launches=72, warm=2, cfs=0.5,
                                        r_comm=0.1, name="c5 (BASELINE configs[4])", key="c5", place_tries=args.leg_place_tries,
                                        cpu=None if args.no_cpu_baseline else dict(budget_all_s=3.0, budget_single_s=1.5)))
        leg("c3", lambda: mappo_iterations(args, args.c3_iters))
        done.set()
    emit()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
