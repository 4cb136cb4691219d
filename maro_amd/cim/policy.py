"""Batched device-side policy pieces of the CIM RL example: the per-port dueling DQN of
``examples/cim/rl/algorithms/dqn.py:13-84`` evaluated for a whole env batch, and the action translation of
``examples/cim/rl/env_sampler.py:33-64`` — so a rollout never leaves the GPU: sampler.state -> q-values -> env actions -> step.

The 22 per-port networks (171 -> 256 -> 128 -> 64 -> 32 -> dueling heads 32 -> 128 -> {21, 1}; LeakyReLU, BatchNorm
in eval mode folded into the affine maps) are stored stacked, ``W[port, in, out]``, and applied as batched GEMMs
(rocBLAS / hipBLASLt MFMA kernels through torch — plain library GEMMs, not part of the simulator hot path).
"""
from __future__ import annotations

from typing import Sequence

import torch

ACTION_SPACE = [(i - 10) / 10 for i in range(21)]   # examples/cim/rl/config.py:20-24


class PerPortDuelingQNet(torch.nn.Module):
    """All ports' Q-networks evaluated at once: q[n, A] of the network that belongs to each env's deciding port."""

    def __init__(self, n_ports: int, state_dim: int, action_num: int = 21, hidden: Sequence[int] = (256, 128, 64, 32),
                 head_hidden: int = 128, dtype=torch.bfloat16, seed: int = 0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        dims = [state_dim] + list(hidden)

        def lin(i, o):
            w = torch.randn((n_ports, i, o), generator=g) * (2.0 / i) ** 0.5
            return torch.nn.Parameter(w.to(dtype)), torch.nn.Parameter(torch.zeros((n_ports, 1, o), dtype=dtype))

        self.trunk = torch.nn.ParameterList([p for i in range(len(hidden)) for p in lin(dims[i], dims[i + 1])])
        self.q1, self.q1b = lin(dims[-1], head_hidden)
        self.q2, self.q2b = lin(head_hidden, action_num)
        self.v1, self.v1b = lin(dims[-1], head_hidden)
        self.v2, self.v2b = lin(head_hidden, 1)
        self.n_ports, self.dtype = n_ports, dtype

    @torch.no_grad()
    def forward(self, states: torch.Tensor, port: torch.Tensor) -> torch.Tensor:
        """states [n, state_dim] (any float dtype), port int [n] -> q-values float32 [n, A] of each env's own port network.
        Every network is evaluated on every state (one batched GEMM per layer; no data-dependent shapes, no host sync) and
        the deciding port's row is gathered."""
        act = torch.nn.functional.leaky_relu
        x = states.to(self.dtype).unsqueeze(0).expand(self.n_ports, -1, -1)
        for i in range(0, len(self.trunk), 2):
            x = act(torch.baddbmm(self.trunk[i + 1], x, self.trunk[i]))
        q = act(torch.baddbmm(self.q2b, act(torch.baddbmm(self.q1b, x, self.q1)), self.q2))       # output_activation LeakyReLU
        v = torch.baddbmm(self.v2b, act(torch.baddbmm(self.v1b, x, self.v1)), self.v2)            # no output activation
        logits = q - q.mean(dim=2, keepdim=True) + v                                             # dqn.py:48-52
        idx = port.to(torch.int64).view(1, -1, 1).expand(1, -1, logits.shape[2])
        return logits.gather(0, idx)[0].float()


def translate_actions(model_action: torch.Tensor, decisions: torch.Tensor, vessel_remaining_space: torch.Tensor,
                      vessel_early_discharge: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """env_sampler.py:33-64 for a batch: model_action int [n] (index into ACTION_SPACE), decisions int32 [n, 8] ->
    engine actions int32 [n, 1, 4] = (vessel_idx, port_idx, quantity, 0=LOAD | 1=DISCHARGE).  Index 10 (value 0.0) is a LOAD
    of 0 containers, as in the reference (zero_action_idx = 10.5).  Arithmetic in float64 with round-half-to-even, like
    Python's round()."""
    n = decisions.shape[0]
    load, discharge = decisions[:, 3].to(torch.float64), decisions[:, 4].to(torch.float64)
    percent = (model_action.to(torch.float64) - 10.0).abs() / 10.0
    is_load = model_action < 10.5
    q_load = torch.minimum(torch.round(percent * load), vessel_remaining_space.to(torch.float64))
    early = vessel_early_discharge.to(torch.float64)
    plan = percent * (discharge + early) - early
    q_dis = torch.where(plan > 0, torch.round(plan), torch.round(percent * discharge))
    qty = torch.where(is_load, q_load, q_dis).to(torch.int32)
    if out is None:
        out = torch.zeros((n, 1, 4), dtype=torch.int32, device=decisions.device)
    out[:, 0, 0] = decisions[:, 2]
    out[:, 0, 1] = decisions[:, 1]
    out[:, 0, 2] = qty
    out[:, 0, 3] = (~is_load).to(torch.int32)
    return out
