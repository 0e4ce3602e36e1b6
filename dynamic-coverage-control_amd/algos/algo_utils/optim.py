"""Adam + gradient-norm clipping on FLAT parameter storage (include/dcc_optim.h).

The reference builds two `torch.optim.Adam` (uav_dcc_control/algos/mappo.py:27-37) and, per PPO step and network, calls
`nn.utils.clip_grad_norm_` then `.step()` (mappo.py:176-185).  Same update rule here, different storage:

  * all parameters of a network are views into ONE flat float32 array (each starting on a 256-byte boundary), their
    `.grad` are views into a second one, the Adam moments are two more;
  * `clip_and_step(max_norm)` is three HIP launches (norm partials, clip coefficient, Adam) instead of ~25 small torch
    launches, with no host synchronisation -- the clip coefficient never leaves the device;
  * `zero_grad()` is one memset, and the multi-GPU path all-reduces `flat_grad` directly (no cat / copy-back).

On a CPU device (the gloo / golden tests of the host logic) the same arithmetic runs as a handful of torch ops on the
flat arrays; on a HIP device the kernels are mandatory (dcc_hip raises if libdcc_hip.so is missing).
"""
import math

import torch

_ALIGN = 64     # floats: every parameter view starts 256-byte aligned (float4 kernels, GEMM operands)


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        params = list(params)
        if not params:
            raise ValueError("FlatAdam: no parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        dev, dt = params[0].device, params[0].dtype
        if dt != torch.float32 or any(p.device != dev or p.dtype != dt for p in params):
            raise ValueError("FlatAdam: float32 parameters on one device")
        self._params = params
        self._numels = [p.numel() for p in params]
        self._offsets, n = [], 0
        for p in params:
            self._offsets.append(n)
            n += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = n
        self.flat_param = torch.zeros(n, dtype=dt, device=dev)
        self.flat_grad = torch.zeros(n, dtype=dt, device=dev)
        self.exp_avg = torch.zeros(n, dtype=dt, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=dt, device=dev)
        with torch.no_grad():
            for p, off in zip(params, self._offsets):
                view = self.flat_param[off:off + p.numel()].view_as(p)
                view.copy_(p)
                p.data = view                                            # the module's parameter now IS the view
                p.grad = self.flat_grad[off:off + p.numel()].view_as(p)   # autograd accumulates in place into the view
        self.step_count = 0
        self._norm_out = torch.zeros(2, dtype=dt, device=dev)           # {grad norm, clip coefficient}
        self._ws = None

    # padding between the views stays zero: zero gradient, zero moments -> Adam leaves it at zero

    def _check_views(self):
        """Every parameter must still BE its view into flat_param: the kernels update flat_param through raw pointers, so a
        parameter whose storage was moved behind the optimizer's back (module.to(...) / any nn.Module._apply -- which also
        makes nn.GRU.flatten_parameters() re-allocate its weights --, load_state_dict(assign=True), p.data = ...) would keep
        computing with a detached tensor while the optimizer trains the orphaned view: training stalls silently.  Host-side
        pointer compare only (no sync).  A moved parameter of unchanged shape / dtype / device is re-attached -- its
        current values, the ones the network computes with, are copied into the view; anything else raises.  Do not call
        .to() on a network after its optimizer was built."""
        base = self.flat_param.data_ptr()
        for p, off, numel in zip(self._params, self._offsets, self._numels):
            if p.data_ptr() == base + 4 * off:
                continue
            view = self.flat_param[off:off + numel]
            if p.device != view.device or p.dtype != view.dtype or p.numel() != numel:
                raise RuntimeError("FlatAdam: a parameter left the optimizer's flat storage and changed device / dtype / size "
                                   "(%s %s, expected %s %s); rebuild the optimizer after moving a network"
                                   % (p.device, p.dtype, view.device, view.dtype))
            with torch.no_grad():
                view.copy_(p.data.reshape(-1))
                p.data = view.view(p.shape)

    def zero_grad(self, set_to_none=False):
        """One memset; the .grad views are re-attached if a caller dropped them (set_to_none elsewhere), and so are
        parameters that left the flat storage (_check_views)."""
        self._check_views()
        self.flat_grad.zero_()
        for p, off in zip(self._params, self._offsets):
            g = p.grad
            if g is None or g.data_ptr() != self.flat_grad.data_ptr() + 4 * off:
                p.grad = self.flat_grad[off:off + p.numel()].view_as(p)

    def _lr(self):
        return float(self.param_groups[0]["lr"])

    @torch.no_grad()
    def clip_and_step(self, max_norm=None):
        """clip_grad_norm_(params, max_norm) (None / <= 0: norm only) followed by one Adam step.  Returns the gradient
        norm BEFORE clipping as a 0-d device tensor (no sync)."""
        self._check_views()
        g = self.param_groups[0]
        beta1, beta2 = g["betas"]
        self.step_count += 1
        t = self.step_count
        step_size = self._lr() / (1.0 - beta1 ** t)
        bc2_sqrt = math.sqrt(1.0 - beta2 ** t)
        mn = float(max_norm) if max_norm else 0.0
        if self.flat_param.is_cuda:
            import dcc_hip
            if self._ws is None:
                self._ws = torch.empty(dcc_hip.grad_norm_workspace_floats(self.numel), dtype=torch.float32, device=self.flat_param.device)
            dcc_hip.grad_norm_clip(self.flat_grad, mn, self._norm_out, self._ws)
            dcc_hip.adam_step(self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, step_size, bc2_sqrt, beta1, beta2,
                              g["eps"], g["weight_decay"], self._norm_out[1:])
            return self._norm_out[0].clone()
        norm = torch.linalg.vector_norm(self.flat_grad)
        clip = torch.clamp(mn / (norm + 1e-6), max=1.0) if mn > 0 else torch.ones((), dtype=norm.dtype)
        gr = self.flat_grad * clip
        if g["weight_decay"] != 0:
            gr = gr + g["weight_decay"] * self.flat_param
        self.exp_avg.lerp_(gr, 1.0 - beta1)
        self.exp_avg_sq.mul_(beta2).addcmul_(gr, gr, value=1.0 - beta2)
        denom = (self.exp_avg_sq.sqrt() / bc2_sqrt).add_(g["eps"])
        self.flat_param.addcdiv_(self.exp_avg, denom, value=-step_size)
        return norm

    def step(self, closure=None):
        self.clip_and_step(None)

    # ---- checkpoints -----------------------------------------------------------------------------------------------------
    def state_dict(self):
        return {"format": "flat_adam_v1", "step": self.step_count, "exp_avg": self.exp_avg.detach().cpu().clone(),
                "exp_avg_sq": self.exp_avg_sq.detach().cpu().clone(), "offsets": list(self._offsets),
                "numels": [p.numel() for p in self._params],
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        if sd.get("format") == "flat_adam_v1":
            if sd["numels"] != [p.numel() for p in self._params]:
                raise ValueError("FlatAdam.load_state_dict: parameter layout differs")
            self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"])
            self.step_count = int(sd["step"])
            for g, sg in zip(self.param_groups, sd["param_groups"]):
                g.update({k: v for k, v in sg.items() if k != "params"})
            return
        # a torch.optim.Adam state_dict (checkpoints written before the flat optimizer): per-parameter moments
        state, groups = sd["state"], sd["param_groups"]
        ids = [i for g in groups for i in g["params"]]
        if len(ids) != len(self._params):
            raise ValueError("FlatAdam.load_state_dict: %d parameters in the checkpoint, %d here" % (len(ids), len(self._params)))
        self.exp_avg.zero_(); self.exp_avg_sq.zero_()
        step = 0
        for i, p, off in zip(ids, self._params, self._offsets):
            st = state.get(i)
            if st is None:
                continue
            self.exp_avg[off:off + p.numel()].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[off:off + p.numel()].copy_(st["exp_avg_sq"].reshape(-1))
            step = max(step, int(float(st["step"])))
        self.step_count = step
        for g, sg in zip(self.param_groups, groups):
            g.update({k: sg[k] for k in ("lr", "betas", "eps", "weight_decay") if k in sg})
