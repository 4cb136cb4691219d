"""``GpuVectorEnv(batch_num, "citi_bike", topology, ...)`` — the VectorEnv / AbsEnv shaped object API
(maro/vector_env/vector_env.py:95-217, simulator/core.py:92-260) over the batched citi_bike engine."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np

from ..cim.vector_env import GpuVectorEnv
from .abi import NODE_ATTRS
from .engine import CitiBikeBatchEngine
from .payloads import encode_action, make_decision_event


class CitiBikeVectorEnv(GpuVectorEnv):
    """``seeds[e]`` plays the role of ``np.random.seed`` in env e's process: the reference draws transfer times
    from the process-global numpy RNG (decision_strategy.py:213-216); reset() rewinds each env's stream, which is
    what a fresh reference process with the same seed would see."""

    ACTION_WIDTH = 3
    METRIC_KEYS = ("trip_requirements", "bike_shortage", "operation_number")
    NODE_ATTRS = NODE_ATTRS

    def __init__(self, batch_num: int, scenario: str = "citi_bike", topology: str = None, start_tick: int = 0,
                 durations: int = 1440, snapshot_resolution: int = 1, max_snapshots: int = None, decision_mode=0,
                 options: Optional[dict] = None, seeds: Optional[Sequence[int]] = None, device="cuda:0", max_actions: int = 4,
                 specialize=None, _engine=None):
        assert scenario == "citi_bike"
        if int(getattr(decision_mode, "value", decision_mode)) != 0:
            raise NotImplementedError("only DecisionMode.Sequential is implemented on the GPU engine")
        if seeds is None:
            seeds = np.arange(batch_num)
        self.engine = _engine if _engine is not None else CitiBikeBatchEngine(
            topology, batch_num, start_tick=start_tick, durations=durations, snapshot_resolution=snapshot_resolution,
            max_snapshots=max_snapshots, max_actions=max_actions, device=device, seeds=seeds, specialize=specialize)
        self._init_state(batch_num)

    def reset(self, keep_seed: bool = False, envs: Optional[Sequence[int]] = None):
        """business_engine.py:164-190: citi_bike has no seed of its own (set_seed is a no-op, :192-193)."""
        envs = list(range(self._n)) if envs is None else list(envs)
        mask = np.zeros(self._n, np.uint8)
        for e in envs:
            mask[e] = 1
            self._started[e] = self._paused[e] = self._finished[e] = False
        self.engine.reset(mask=mask)

    def set_seed(self, seed: int, envs: Optional[Sequence[int]] = None):
        pass

    # ---- scenario hooks
    def _encode_action(self, a) -> tuple:
        return encode_action(a)

    def _engine_step(self, acts, nact, mask, n_answered=None):
        dec, scope, met, done = self.engine.step(acts, nact, mask)
        return dec.cpu().numpy(), met.cpu().numpy(), done.cpu().numpy(), scope.cpu().numpy()

    def _make_event(self, e: int, row, extra):
        return make_decision_event(row, extra[e])

    def _node_counts(self) -> Dict[str, int]:
        return {"stations": self.engine.data.n_stations, "matrices": 1}

    def _ring_fi_rows(self) -> np.ndarray:
        return self.engine.ring_fi.cpu().numpy().T

    def _agent_idx_list(self) -> List[int]:
        return list(range(self.engine.data.n_stations))

    def _configs(self) -> dict:
        d = self.engine.data
        return {"decision": {"resolution": d.resolution, "supply_water_mark_ratio": d.supply_water_mark_ratio,
                             "demand_water_mark_ratio": d.demand_water_mark_ratio}}

    def _name(self) -> str:
        return f"citi_bike:{self.engine.data.name}"

    def _summary(self) -> dict:
        d = self.engine.data
        return {"node_mapping": {i: int(x) for i, x in enumerate(d.station_id)},
                "node_detail": {"stations": {"number": d.n_stations}, "matrices": {"number": 1}}, "event_payload": {}}
