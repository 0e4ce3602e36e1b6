"""Stand-alone PyTorch-only probe (no code of this repo): large multi-block reductions inside a captured-and-replayed
hipGraph return wrong values on this stack (ROCm 7.2 / PyTorch 2.10+rocm7.0, MI355X) after a few replays.
This is why whole PPO epochs (full of such reductions) are NOT replayed as graphs -- the round-1 experiment
`use_hip_graph_update` was removed in round 2 -- and why the rollout graph (`use_hip_graph`) accumulates its logged
statistics element-wise per env and reduces them after the replay; it contains no torch reduction and is verified bit-identical to eager
(tools/graph_rollout_soak.py).  Observations: every full-tensor reduction flavour (sum, mean, vector_norm, mean of squares) of
>= 131 k elements fails, from the same replay index on (~815 with 24 reductions per replay) and then persistently; row-wise
reductions ([600, 1024].sum(1)) and reductions of <= 32 k elements never do; a single such reduction per replay with changing
inputs is right for the first replays -- i.e. the semaphore-reset memset of PyTorch's multi-block reduce is executed at first,
and something in the replay machinery gives out after some tens of thousands of replayed nodes of that kind."""
import sys
import torch

dev = "cuda"
torch.manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 614400
interleave = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
x = torch.randn(n, 1, device=dev) * 2000 - 2000


def body(x):
    outs = []
    for k in range(6):
        outs.append(x.mean(dim=(0,)))
        outs.append((x ** 2).mean(dim=(0,)))
        outs.append((x * float(k + 1)).sum())
        outs.append(torch.linalg.vector_norm(x))
    return torch.stack([o.reshape(()) for o in outs])


ref = body(x).clone()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = body(x)
bad, first = 0, None
for i in range(2000):
    if interleave:      # eager allocations / kernels between replays, like a training loop
        y = torch.empty(1 << 20, device=dev).normal_()
    g.replay()
    if not torch.equal(out, ref):
        bad += 1
        first = i if first is None else first
torch.cuda.synchronize()
print("n=%d interleaved_eager_work=%s: 2000 replays, %d mismatching (first at replay %s)" % (n, interleave, bad, first))
