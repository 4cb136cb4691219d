"""Batched, on-device observation and reward shaping for the CIM RL example.

Mirrors ``examples/cim/rl/env_sampler.py:15-31`` (state) and ``:65-80`` (delayed reward) with the shaping
constants of ``examples/cim/rl/config.py:13-29``, for every env of a ``CimBatchEngine`` at once: three snapshot
queries and a handful of tensor ops, no per-env Python loop and no host round trip.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

PORT_ATTRIBUTES = ["empty", "full", "on_shipper", "on_consignee", "booking", "shortage", "fulfillment"]
VESSEL_ATTRIBUTES = ["empty", "full", "remaining_space"]


class CimBatchSampler:
    def __init__(self, engine, look_back: int = 7, time_window: int = 99, fulfillment_factor: float = 1.0,
                 shortage_factor: float = 1.0, time_decay: float = 0.97, port_attributes: Sequence[str] = PORT_ATTRIBUTES,
                 vessel_attributes: Sequence[str] = VESSEL_ATTRIBUTES):
        self.eng = engine
        self.look_back, self.time_window = look_back, time_window
        self.ff, self.sf = fulfillment_factor, shortage_factor
        self.port_attributes, self.vessel_attributes = list(port_attributes), list(vessel_attributes)
        dev = engine.decisions.device
        self._back = torch.arange(look_back - 1, dtype=torch.int32, device=dev)          # range(look_back - 1)
        self._ahead = torch.arange(1, time_window + 1, dtype=torch.int32, device=dev)    # tick + 1 ... tick + window
        self._decay = torch.tensor([time_decay ** i for i in range(time_window)], dtype=torch.float64, device=dev)
        self.state_dim = (look_back - 1) * (1 + engine.topo.future_stop_number) * len(self.port_attributes) + len(
            self.vessel_attributes)

    def state(self, decisions: Optional[torch.Tensor] = None) -> torch.Tensor:
        """float64 [n_envs, state_dim] for the pending decision of every env (rows of finished envs are meaningless).
        env_sampler.py:21-31: ports = [port] + future_stop_list of the vessel; ticks = max(0, tick - rt)."""
        d = self.eng.decisions if decisions is None else decisions
        n = d.shape[0]
        tick, port, vessel = d[:, 0:1], d[:, 1:2], d[:, 2:3]
        fut = self.eng.query("vessels", tick, vessel, ["future_stop_list"]).view(n, -1).to(torch.int32)
        nodes = torch.cat([port.to(torch.int32), fut], dim=1).contiguous()
        ticks = torch.clamp(tick - self._back[None, :], min=0).to(torch.int32).contiguous()
        ps = self.eng.query("ports", ticks, nodes, self.port_attributes).view(n, -1)
        vs = self.eng.query("vessels", tick, vessel, self.vessel_attributes).view(n, -1)
        return torch.cat([ps, vs], dim=1)

    def reward(self, tick: torch.Tensor, port: torch.Tensor) -> torch.Tensor:
        """float32 [n_envs]: sum_i decay^i * (ff * fulfillment - sf * shortage) of `port` over ticks tick+1 .. tick+window
        (env_sampler.py:65-80).  Frames the ring does not hold contribute zeros, like the reference's padding."""
        n = tick.shape[0]
        ticks = (tick.view(n, 1).to(torch.int32) + self._ahead[None, :]).contiguous()
        nodes = port.view(n, 1).to(torch.int32).contiguous()
        q = self.eng.query("ports", ticks, nodes, ["fulfillment", "shortage"]).view(n, self.time_window, 2)
        r = self.ff * (q[:, :, 0] @ self._decay) - self.sf * (q[:, :, 1] @ self._decay)
        return r.to(torch.float32)
