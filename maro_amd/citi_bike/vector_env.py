"""``GpuVectorEnv(batch_num, "citi_bike", topology, ...)`` — the VectorEnv / AbsEnv shaped object API
(maro/vector_env/vector_env.py:95-217, simulator/core.py:92-260) over the batched citi_bike engine."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np

from ..cim.vector_env import GpuVectorEnv, _as_list
from .abi import NODE_ATTRS
from .engine import CitiBikeBatchEngine
from .payloads import encode_action, make_decision_event


class CitiBikeVectorEnv(GpuVectorEnv):
    """``seeds[e]`` plays the role of ``np.random.seed`` in env e's process: the reference draws transfer times
    from the process-global numpy RNG (decision_strategy.py:213-216); reset() rewinds each env's stream, which is
    what a fresh reference process with the same seed would see."""

    ACTION_WIDTH = 3
    _WHOLE_BATCH_STEP = False   # (events carry per-env scope rows: the per-env path builds them)
    METRIC_KEYS = ("trip_requirements", "bike_shortage", "operation_number")
    NODE_ATTRS = NODE_ATTRS

    def __init__(self, batch_num: int, scenario: str = "citi_bike", topology: str = None, start_tick: int = 0,
                 durations: int = 1440, snapshot_resolution: int = 1, max_snapshots: int = None, decision_mode=0,
                 options: Optional[dict] = None, seeds: Optional[Sequence[int]] = None, device="cuda:0", max_actions: int = 4,
                 specialize=None, _engine=None):
        assert scenario == "citi_bike"
        mode = int(getattr(decision_mode, "value", decision_mode))
        if mode not in (0, 1, 2):
            raise ValueError("decision_mode must be Sequential (0), Joint (1) or JointWithSequentialAction (2)")
        if seeds is None:
            seeds = np.arange(batch_num)
        self.engine = _engine if _engine is not None else CitiBikeBatchEngine(
            topology, batch_num, start_tick=start_tick, durations=durations, snapshot_resolution=snapshot_resolution,
            max_snapshots=max_snapshots, max_actions=max_actions, device=device, seeds=seeds, specialize=specialize, decision_mode=mode)
        self._mode = int(getattr(self.engine, "decision_mode", mode))
        self._init_state(batch_num)

    def reset(self, keep_seed: bool = False, envs: Optional[Sequence[int]] = None):
        """business_engine.py:164-190: citi_bike has no seed of its own (set_seed is a no-op, :192-193)."""
        envs = list(range(self._n)) if envs is None else list(envs)
        mask = np.zeros(self._n, np.uint8)
        for e in envs:
            mask[e] = 1
            self._started[e] = self._paused[e] = self._finished[e] = False
            self._joint_events.pop(e, None)
            self._last_met[e] = 0
            self._n_pending[e] = 0
            self._last_dec[e] = None
        self._host_ticks = None
        self.engine.reset(mask=mask)

    def set_seed(self, seed: int, envs: Optional[Sequence[int]] = None):
        pass

    # ---- Joint / JointWithSequentialAction (core.py:354-366): mrx_cb_step_joint takes one action list per reported event
    def _step_envs(self, envs: Sequence[int], per_env: Dict[int, object]):
        if self._mode == 0:
            return super()._step_envs(envs, per_env)
        eng = self.engine
        A, S = eng.max_actions, eng.data.n_stations
        acts = np.full((self._n, S, A, 3), -1, np.int32)
        nact = np.zeros((self._n, S), np.int32)
        nans = np.zeros(self._n, np.int32)
        mask = np.zeros(self._n, np.uint8)
        out = {}
        for e in envs:
            if self._finished[e]:
                out[e] = (None, None, True)          # core.py:128-133
                continue
            mask[e] = 1
            per_event = per_env.get(e)
            per_event = [] if per_event is None else list(per_event) if isinstance(per_event, (list, tuple)) else [per_event]
            per_event = per_event[:int(self._n_pending[e])]   # zip(actions, pending_events)
            nans[e] = len(per_event)
            for i, entry in enumerate(per_event):
                alist = _as_list(entry)
                if len(alist) > A:
                    raise ValueError(f"{len(alist)} actions for one decision event; engine was built with max_actions={A}")
                for j, a in enumerate(alist):
                    acts[e, i, j] = self._encode_action(a)
                nact[e, i] = len(alist)
        if mask.any():
            dec, scope, met, done = (x.cpu().numpy() for x in eng.step_joint(acts, nact, nans, mask))
            self._host_ticks = None
            for e in envs:
                if not mask[e]:
                    continue
                self._started[e] = True
                self._last_dec[e], self._last_met[e] = dec[e, 0], met[e]
                metrics = {k: int(met[e, i]) for i, k in enumerate(self.METRIC_KEYS)}
                if done[e]:
                    self._paused[e], self._finished[e] = False, True
                    self._joint_events.pop(e, None)
                    out[e] = (metrics, None, True)
                    continue
                self._paused[e] = True
                # an event that stayed pending (JointWithSequentialAction) is re-yielded as the SAME object, its action scope
                # cached at the first read (citi_bike/common.py:96-104), like the reference does
                cache = self._joint_events.get(e, {})
                events, keep = [], {}
                for k in range(int(dec[e, 0, 6])):
                    key = (int(dec[e, k, 0]), int(dec[e, k, 1]))
                    ev = cache.get(key) or make_decision_event(dec[e, k], scope[e, k])
                    keep[key] = ev
                    events.append(ev)
                self._joint_events[e] = keep
                self._n_pending[e] = len(events)
                out[e] = (metrics, events, False)
        return [out[e] for e in envs]

    # ---- scenario hooks
    def _encode_action(self, a) -> tuple:
        return encode_action(a)

    def _engine_step(self, acts, nact, mask, n_answered=None):
        dec, scope, met, done = self.engine.step(acts, nact, mask)
        self._host_ticks = None
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
