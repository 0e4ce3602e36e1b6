"""GRU + LayerNorm layer of the recurrent policy variants (reference: uav_dcc_control/algos/algo_utils/rnn.py:7-82;
used by R_Actor / R_Critic when `use_recurrent_policy` or `use_naive_recurrent_policy` is set -- both are off in
the shipped mappo.yaml).  Same parameters and names (`rnn.weight_ih_l0`, ..., `norm.weight`) and the same function:

  * one step (rollout):   x [B,H], hxs [B,recurrent_N,H], masks [B,1]  ->  GRU(x, hxs * mask)
  * a sequence (update):  x [T*B,H] (time-major flattening), hxs [B,recurrent_N,H] = state at t=0, masks [T*B,1]

The reference cuts the sequence at the steps where any mask is 0 and runs one multi-step GRU call per segment; to find
the cut points it copies the masks to the host (`.nonzero().cpu()`: a device synchronisation inside every PPO
mini-batch).  Multiplying the state by the step's mask before EVERY step is the same arithmetic (the mask is exactly
1.0 inside a segment) and needs no host round trip, so the update stays asynchronous on the device.
"""
import torch
import torch.nn as nn


class RNNLayer(nn.Module):
    def __init__(self, inputs_dim, outputs_dim, recurrent_N, use_orthogonal):
        super().__init__()
        self._recurrent_N = recurrent_N
        self.rnn = nn.GRU(inputs_dim, outputs_dim, num_layers=recurrent_N)
        for name, param in self.rnn.named_parameters():   # rnn.py:15-22
            if "bias" in name:
                nn.init.constant_(param, 0)
            elif "weight" in name:
                (nn.init.orthogonal_ if use_orthogonal else nn.init.xavier_uniform_)(param)
        self.norm = nn.LayerNorm(outputs_dim)

    def forward(self, x, hxs, masks):
        B = hxs.size(0)
        h = hxs.transpose(0, 1).contiguous()                       # [recurrent_N, B, H]
        if x.size(0) == B:
            out, h = self.rnn(x.unsqueeze(0), h * masks.view(1, B, 1))
            x = out.squeeze(0)
        else:
            T = x.size(0) // B
            xs, ms = x.view(T, B, -1), masks.view(T, 1, B, 1)
            outs = []
            for t in range(T):
                out, h = self.rnn(xs[t:t + 1], h * ms[t])
                outs.append(out)
            x = torch.cat(outs, dim=0).reshape(T * B, -1)
        return self.norm(x), h.transpose(0, 1)
