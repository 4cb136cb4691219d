"""Reference-shaped object API over the batched engine.

``GpuVectorEnv`` mirrors ``maro.vector_env.VectorEnv`` (vector_env.py:95-217): ``step(action)`` with a
scalar / list / dict action, ``reset()``, ``snapshot_list[node][ticks:nodes:attrs]`` (a list of flat float64
numpy arrays, one per env), ``tick``, ``frame_index``.  ``env_view(i)`` gives an ``AbsEnv``-shaped single-env
handle (core.py:92-260) so code written against ``Env`` — e.g. a ``maro.rl`` ``AbsEnvSampler`` — can drive
env ``i`` of the batch unchanged.  This layer creates Python objects per env and synchronises with the
device every call; high-throughput rollouts use the tensor API of ``CimBatchEngine`` directly.
"""
from __future__ import annotations

import gc
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .engine import NODE_ATTRS, SEED_KEEP, SEED_REDRAW, CimBatchEngine
from .payloads import _ACTION_CODE, HAVE_MARO, ActionScope, ActionType, DecisionEvent, action_code, encode_action, make_decision_event

try:   # the per-env object loops of a whole-batch step in C (maro_amd/csrc_host/fastobj.c, built by __graft_entry__.build()); optional
    from .. import _fastobj as _FO
except ImportError:   # not built: the comprehensions below do the same work
    _FO = None


class InvalidActionError(AssertionError):
    """The reference asserts inside ``_on_action_received`` (cim/business_engine.py:731,736)."""


class BackendsInvalidAttributeException(Exception):
    """Error code 2110 of the reference (maro/utils/exception/backends_exception.py)."""

    def __init__(self, msg="Attribute not exist"):
        super().__init__(2110, msg)


def _as_list(x) -> list:
    if x is None:
        return []
    if isinstance(x, (list, tuple)):
        return list(x)
    return [x]


class object_api_gc:
    """``with object_api_gc(batch_num): ...`` around a rollout through the OBJECT API.  A batch step hands out tens of thousands
    of small Python objects (DecisionEvents, metrics dicts) and the agent answers with as many Actions; with CPython's default
    collector thresholds every 700th allocation walks the young generation and the old generations follow — measured at 16 384
    envs: 87 ms of a 116 ms step, against 8 ms with the collector out of the way.  Entering freezes what is alive
    (``gc.freeze()``) and lifts the young-generation threshold above one step's allocations; leaving restores both.  Nothing in
    the engine depends on it — it is the two lines a user would otherwise write."""

    def __init__(self, batch_num: int):
        self._n = int(batch_num)

    def __enter__(self):
        self._thr = gc.get_threshold()
        gc.collect()
        gc.freeze()
        gc.set_threshold(max(self._thr[0], 8 * self._n), self._thr[1], self._thr[2])
        return self

    def __exit__(self, *a):
        gc.set_threshold(*self._thr)
        gc.unfreeze()


class _SnapshotNode:
    """``snapshot_list[node]`` — slice grammar of frame.pyx:754-801."""

    def __init__(self, owner: "GpuVectorEnv", node: str, envs: Optional[Sequence[int]]):
        self._o, self._node, self._envs = owner, node, envs

    def __len__(self):
        return self._o._node_counts()[self._node]

    def __getitem__(self, key: slice):
        if key.step is None:  # querying needs at least one attribute
            return None
        attrs = _as_list(key.step)
        for a in attrs:
            if a not in self._o.NODE_ATTRS[self._node]:
                raise BackendsInvalidAttributeException()
        out = self._o._query(self._node, _as_list(key.start), _as_list(key.stop), attrs)
        if self._envs is None:
            return out
        return out[self._envs[0]] if len(self._envs) == 1 else [out[e] for e in self._envs]


class _SnapshotList:
    def __init__(self, owner: "GpuVectorEnv", envs: Optional[Sequence[int]] = None):
        self._o, self._envs = owner, envs

    def __getitem__(self, name: str):
        return _SnapshotNode(self._o, name, self._envs) if name in self._o.NODE_ATTRS else None

    def get_frame_index_list(self):
        lists = self._o._frame_index_lists()
        if self._envs is None:
            return lists
        return lists[self._envs[0]] if len(self._envs) == 1 else [lists[e] for e in self._envs]

    def __len__(self):
        lists = self._o._frame_index_lists()
        return len(lists[self._envs[0]]) if self._envs else max(len(x) for x in lists)


class GpuVectorEnv:
    """``batch_num`` environments on one GPU, VectorEnv-shaped.  ``scenario`` "cim" is implemented here;
    "citi_bike" dispatches to ``maro_amd.citi_bike.vector_env.CitiBikeVectorEnv`` (same surface)."""

    ACTION_WIDTH = 4
    _WHOLE_BATCH_STEP = True    # step(None | Action | list) of the whole batch takes the array-form path (_step_all)
    NODE_ATTRS = NODE_ATTRS
    METRIC_KEYS = ("order_requirements", "container_shortage", "operation_number")

    def __new__(cls, batch_num: int = 0, scenario: str = "cim", *args, **kwargs):
        if cls is GpuVectorEnv and scenario == "citi_bike":
            from ..citi_bike.vector_env import CitiBikeVectorEnv
            return super().__new__(CitiBikeVectorEnv)
        if cls is GpuVectorEnv and (scenario not in ("cim", "citi_bike") or kwargs.get("business_engine_cls") is not None):
            # SURVEY.md 8(b) "custom BE plugin": scenarios / business engines the GPU engines do not implement run on the
            # reference's own process-per-env VectorEnv when MARO is importable (core.py:51, 270-271); there is no GPU path
            # and nothing is emulated here.
            try:
                from maro.vector_env import VectorEnv
            except ImportError as e:
                raise NotImplementedError(
                    f"the GPU engines implement the 'cim' and 'citi_bike' scenarios; scenario={scenario!r} / a custom "
                    f"business_engine_cls needs the reference (maro.vector_env.VectorEnv), which is not importable: {e}") from None
            # bind the positional arguments by name exactly as VectorEnv.__init__ declares them (vector_env.py:62-72): a
            # positional topology must not land in the `scenario` slot, and the GPU-only keywords are dropped
            names = ("topology", "start_tick", "durations", "snapshot_resolution", "max_snapshots", "decision_mode",
                     "business_engine_cls", "disable_finished_events", "options")
            if len(args) > len(names):
                raise TypeError(f"GpuVectorEnv takes at most {len(names) + 2} positional arguments")
            bound = dict(zip(names, args))
            for k, v in kwargs.items():
                if k in bound:
                    raise TypeError(f"GpuVectorEnv got multiple values for argument {k!r}")
                if k not in ("seeds", "device", "max_actions", "specialize", "_engine"):
                    bound[k] = v
            if "decision_mode" in bound and isinstance(bound["decision_mode"], int) and not isinstance(bound["decision_mode"], bool):
                from maro.simulator.utils.common import DecisionMode   # the GPU path takes plain ints, the reference wants the enum
                bound["decision_mode"] = {0: DecisionMode.Sequential, 1: DecisionMode.Joint, 2: DecisionMode.JointWithSequentialAction}[bound["decision_mode"]]
            return VectorEnv(batch_num, scenario=scenario, **bound)
        return super().__new__(cls)

    def __init__(self, batch_num: int, scenario: str = "cim", topology: str = None, start_tick: int = 0,
                 durations: int = 100, snapshot_resolution: int = 1, max_snapshots: int = None, decision_mode=0,
                 options: Optional[dict] = None, seeds: Optional[Sequence[int]] = None, device="cuda:0",
                 max_actions: int = 4, specialize=None, groups: int = 1, _engine=None):
        """specialize: passed to the engine (CimBatchEngine / CitiBikeBatchEngine): True = kernels compiled for this plan.
        groups > 1 (CIM): the batch runs as that many independent engines on their own HIP streams (PipelinedCimBatch,
        DESIGN.md section 2: the other groups' kernels fill a step kernel's tail) — same results, env order unchanged."""
        if scenario != "cim":
            raise NotImplementedError("the GPU engine implements the 'cim' and 'citi_bike' scenarios; use "
                                      "maro.simulator.Env for others")
        mode = int(getattr(decision_mode, "value", decision_mode))
        if mode not in (0, 1, 2):
            raise ValueError("decision_mode must be Sequential (0), Joint (1) or JointWithSequentialAction (2)")
        ekw = dict(start_tick=start_tick, durations=durations, snapshot_resolution=snapshot_resolution, max_snapshots=max_snapshots,
                   max_actions=max_actions, seeds=seeds, decision_mode=mode, specialize=specialize)
        if _engine is not None:
            self.engine = _engine
        elif int(groups) > 1:
            from .rollout import PipelinedCimBatch
            ekw["seeds"] = "topology" if seeds is None else seeds
            self.engine = PipelinedCimBatch(topology, batch_num, groups=int(groups), device=device, **ekw)
        else:
            self.engine = CimBatchEngine(topology, batch_num, device=device, **ekw)
        self._mode = int(getattr(self.engine, "decision_mode", mode))
        self._init_state(batch_num)

    def _init_state(self, batch_num: int):
        self._n = batch_num
        self._started = np.zeros(batch_num, bool)
        self._paused = np.zeros(batch_num, bool)
        self._finished = np.zeros(batch_num, bool)     # (metrics, None, True) already returned
        self._pending_seed: Dict[int, int] = {}
        self._last_dec = [None] * batch_num
        self._joint_events: Dict[int, dict] = {}   # env -> {(tick, vessel): DecisionEvent} (JointWithSequentialAction)
        self._n_pending = np.zeros(batch_num, np.int32)
        self._last_met = np.zeros((batch_num, 3), np.int64)
        self._snapshots = _SnapshotList(self)
        self._host_ticks = None                        # ticks read back with the last step's results (None: stale)
        self._env_snapshots = [None] * batch_num       # per-env snapshot-list views handed to decision events (made on first use)

    # ------------------------------------------------------------------ VectorEnv surface
    @property
    def batch_number(self) -> int:
        return self._n

    @property
    def snapshot_list(self) -> _SnapshotList:
        return self._snapshots

    def _ticks_host(self) -> np.ndarray:
        """The envs' current ticks on the host: the copy taken with the last step's results (one read-back per step, however many
        `tick` / `frame_index` reads follow), re-read after a reset."""
        if self._host_ticks is None:
            self._host_ticks = self.engine.ticks.cpu().numpy()
        return self._host_ticks

    @property
    def tick(self) -> List[int]:
        return self._ticks_host().tolist()

    @property
    def frame_index(self) -> List[int]:
        e = self.engine
        return ((self._ticks_host() - e.start_tick) // e.snapshot_resolution).tolist()

    def step(self, action=None):
        """vector_env.py:116-144.  Returns (metrics list, decision_event list, all_done)."""
        if isinstance(action, dict):
            envs = sorted(action.keys())
            per_env = {e: action[e] for e in envs}
        elif isinstance(action, list):
            assert len(action) == self._n
            envs = list(range(self._n))
            per_env = dict(enumerate(action))
        else:
            envs = list(range(self._n))
            per_env = {e: action for e in envs}
        if self._WHOLE_BATCH_STEP and getattr(self, "_mode", 0) == 0 and not isinstance(action, dict):
            return self._step_all(action)
        res = self._step_envs(envs, per_env)
        return [r[0] for r in res], [r[1] for r in res], bool(self._finished.all())

    def _step_all(self, action):
        """``step`` of the WHOLE batch in Sequential mode — the same results as ``_step_envs`` over every env, with the per-env
        work in whole-array form: actions encoded in one pass, ONE read-back of the step's results (decisions, metrics, done,
        status, ticks), bookkeeping as array operations, the metrics dicts and DecisionEvents built by list comprehensions over
        plain Python ints (what is left per env is the objects the reference's API promises)."""
        eng, n, A = self.engine, self._n, self.engine.max_actions
        fin_before = self._finished.copy()
        mask = (~fin_before).astype(np.uint8)
        acts = np.zeros((n, A, self.ACTION_WIDTH), np.int32)
        nact = np.zeros(n, np.int32)
        if action is not None:
            if isinstance(action, list) and _FO is not None:
                multi = _FO.encode_actions(action, fin_before.view(np.uint8), acts, nact, A, ActionType.LOAD, ActionType.DISCHARGE)
                for e, a in multi:
                    if len(a) > A:
                        raise ValueError(f"{len(a)} actions for one decision event; engine was built with max_actions={A}")
                    for i, x in enumerate(a):
                        acts[e, i] = self._encode_action(x)
                    nact[e] = len(a)
            elif isinstance(action, list):
                rows, idx, multi = self._encode_rows(action, fin_before.tolist())
                for e, a in multi:
                    if len(a) > A:
                        raise ValueError(f"{len(a)} actions for one decision event; engine was built with max_actions={A}")
                    for i, x in enumerate(a):
                        acts[e, i] = self._encode_action(x)
                    nact[e] = len(a)
                if idx:
                    acts[idx, 0] = np.asarray(rows, np.int32)
                    nact[idx] = 1
            else:
                one = _as_list(action)
                if len(one) > A:
                    raise ValueError(f"{len(one)} actions for one decision event; engine was built with max_actions={A}")
                for i, x in enumerate(one):
                    acts[:, i] = self._encode_action(x)
                nact[:] = len(one)
                nact[fin_before] = 0
        metrics, events = [None] * n, [None] * n
        if mask.any():
            dec, met, done, extra = self._engine_step(acts, nact, mask, None)
            status = eng.status.cpu().numpy()
            self._host_ticks = eng.ticks.cpu().numpy()
            live = mask.astype(bool)
            done = done.astype(bool)
            self._started |= live
            self._last_dec = dec
            self._last_met[live] = met[live]
            self._paused = np.where(live, ~done, self._paused)
            self._finished |= live & done
            k0, k1, k2 = self.METRIC_KEYS
            met_l, live_l = (None, None) if _FO is not None else (met.tolist(), live.tolist())
            # tens of thousands of small acyclic objects at once: with the cyclic collector on, every 700th allocation walks
            # the young generation (measured: 3.5 x the construction time at 16384 envs), so it is paused for the build
            gc_was_on = gc.isenabled()
            gc.disable()
            try:
                if _FO is not None:
                    metrics = _FO.build_metrics(np.ascontiguousarray(met, np.int64), live.view(np.uint8), k0, k1, k2)
                else:
                    metrics = [{k0: m[0], k1: m[1], k2: m[2]} if lv else None for m, lv in zip(met_l, live_l)]
                events = self._make_events(dec, extra, live & ~done)
            finally:
                if gc_was_on:
                    gc.enable()
            offenders = np.flatnonzero(live & ((status & 1) != 0)).tolist()
            if offenders:   # (see _step_envs: the whole batch has stepped, the offenders' INVALID_ACTION bits are cleared, all are named)
                if hasattr(eng, "clear_status_bits"):
                    eng.clear_status_bits(offenders, 1)
                else:
                    idx = torch.as_tensor(offenders, dtype=torch.int64, device=eng.status.device)
                    eng.status[idx] = eng.status[idx] & ~1
                raise InvalidActionError(f"env(s) {offenders}: invalid action (cim/business_engine.py:731,736); the action was skipped")
        metrics = [(None if fb else m) for m, fb in zip(metrics, fin_before.tolist())] if fin_before.any() else metrics
        return metrics, events, bool(self._finished.all())

    def _env_snapshot_list(self, e: int):
        sl = self._env_snapshots[e]
        if sl is None:
            sl = self._env_snapshots[e] = _SnapshotList(self, [e])
        return sl

    def _make_events(self, dec, extra, want) -> list:
        """DecisionEvents of the envs flagged in `want` (bool array [n]) from the step's decision rows."""
        snaps = self._env_snapshots
        if None in snaps:
            snaps = self._env_snapshots = [sl if sl is not None else _SnapshotList(self, [e]) for e, sl in enumerate(snaps)]
        if _FO is not None and not HAVE_MARO:
            d32 = np.ascontiguousarray(dec, np.int32)
            return _FO.build_events(DecisionEvent, d32, int(d32.shape[1]), np.ascontiguousarray(want).view(np.uint8), snaps, ActionScope)
        rows = dec[:, :6].tolist()
        want = want.tolist()
        if HAVE_MARO:
            return [make_decision_event(r, sl) if w else None for r, sl, w in zip(rows, snaps, want)]
        # (DecisionEvent._from_row spelled out in the loop: one object + one dict literal per event, no call)
        out, cls = [], DecisionEvent
        app, new = out.append, DecisionEvent.__new__
        for r, sl, w in zip(rows, snaps, want):
            if w:
                ev = new(cls)
                ev.__dict__ = {"tick": r[0], "port_idx": r[1], "vessel_idx": r[2], "snapshot_list": sl, "_early_discharge": r[5], "_scope": (r[3], r[4])}
                app(ev)
            else:
                app(None)
        return out

    def _encode_rows(self, actions, skip) -> tuple:
        """(rows, env indices) of the entries of `actions` that are ONE Action object; None entries and the envs in `skip` are
        left out, list / tuple entries are returned as (env, entry) pairs for the caller's general path."""
        rows, idx, multi = [], [], []
        code = _ACTION_CODE.get
        for e, a in enumerate(actions):
            if a is None or skip[e]:
                continue
            if isinstance(a, (list, tuple)):
                multi.append((e, a))
                continue
            c = code(a.action_type)
            rows.append((a.vessel_idx, a.port_idx, a.quantity, c if c is not None else action_code(a.action_type)))
            idx.append(e)
        return rows, idx, multi

    def reset(self, keep_seed: bool = False, envs: Optional[Sequence[int]] = None):
        """VectorEnv.reset resets every env with Env.reset() (keep_seed=False, env_process.py:43-50)."""
        envs = list(range(self._n)) if envs is None else list(envs)
        cmd = np.full(self._n, SEED_KEEP, np.int64)
        mask = np.zeros(self._n, np.uint8)
        for e in envs:
            mask[e] = 1
            if not keep_seed:
                cmd[e] = SEED_REDRAW       # set_seed is overridden by the redraw (cim_data_container_helpers.py:58-60)
            elif e in self._pending_seed:
                cmd[e] = self._pending_seed[e]
            self._pending_seed.pop(e, None)
            self._started[e] = self._paused[e] = self._finished[e] = False
            self._joint_events.pop(e, None)
            self._last_met[e] = 0          # Env.reset: fresh metrics (business_engine.py:226-242)
            self._n_pending[e] = 0
            if isinstance(self._last_dec, list):
                self._last_dec[e] = None
        self._host_ticks = None
        self.engine.reset(cmd, mask)

    def set_seed(self, seed: int, envs: Optional[Sequence[int]] = None):
        """Takes effect at the next reset (cim_data_container_helpers.py:68-70)."""
        for e in (range(self._n) if envs is None else envs):
            self._pending_seed[e] = int(seed)

    def env_view(self, index: int) -> "GpuEnvView":
        return GpuEnvView(self, index)

    def stop(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.stop()

    # ------------------------------------------------------------------ internals
    def _step_envs(self, envs: Sequence[int], per_env: Dict[int, object]):
        eng = self.engine
        A = eng.max_actions
        acts = np.zeros((self._n, A, self.ACTION_WIDTH), np.int32)
        nact = np.zeros(self._n, np.int32)
        nans = np.full(self._n, -1, np.int32)
        mask = np.zeros(self._n, np.uint8)
        joint = getattr(self, "_mode", 0) != 0
        out = {}
        for e in envs:
            if self._finished[e]:
                out[e] = (None, None, True)          # core.py:128-133
                continue
            mask[e] = 1
            if joint:
                # core.py:354-366: one entry (Action / list / None) per pending event, zipped in event order
                per_event = per_env.get(e)
                per_event = [] if per_event is None else list(per_event) if isinstance(per_event, (list, tuple)) else [per_event]
                per_event = per_event[:int(self._n_pending[e])]
                nans[e] = len(per_event)
                alist = [a for entry in per_event for a in _as_list(entry)]
            else:
                alist = _as_list(per_env.get(e))
            if len(alist) > A:
                raise ValueError(f"{len(alist)} actions for one decision event; engine was built with max_actions={A}")
            for i, a in enumerate(alist):
                acts[e, i] = self._encode_action(a)
            nact[e] = len(alist)
        if mask.any():
            dec, met, done, extra = self._engine_step(acts, nact, mask, nans if joint else None)
            status = eng.status.cpu().numpy()
            self._host_ticks = None
            offenders = [e for e in envs if mask[e] and (status[e] & 1)]
            for e in envs:
                if not mask[e]:
                    continue
                self._started[e] = True
                if not isinstance(self._last_dec, list):
                    self._last_dec = list(self._last_dec)
                self._last_dec[e], self._last_met[e] = dec[e], met[e]
                metrics = {k: int(met[e, i]) for i, k in enumerate(self.METRIC_KEYS)}
                if done[e]:
                    self._paused[e], self._finished[e] = False, True
                    out[e] = (metrics, None, True)
                else:
                    self._paused[e] = True
                    out[e] = (metrics, self._make_joint_events(e, dec[e]) if joint else self._make_event(e, dec[e], extra), False)
            if offenders:
                # The whole masked batch has stepped (the engine skips an action the reference would reject with an
                # AssertionError, cim/business_engine.py:731,736, and carries on); the bookkeeping above is complete for
                # every env, only the INVALID_ACTION bit of the offenders is cleared, and all of them are named.
                if hasattr(eng, "clear_status_bits"):
                    eng.clear_status_bits(offenders, 1)
                else:
                    idx = torch.as_tensor(offenders, dtype=torch.int64, device=eng.status.device)
                    eng.status[idx] = eng.status[idx] & ~1
                raise InvalidActionError(f"env(s) {offenders}: invalid action (cim/business_engine.py:731,736); the action was skipped")
        return [out[e] for e in envs]

    def _make_joint_events(self, e: int, rows) -> list:
        """Joint modes: every pending event of the tick.  An event that stayed pending is re-yielded as the SAME object
        (its action scope cached at the first read, cim/common.py:107-123), like the reference does."""
        cache = self._joint_events.setdefault(e, {})
        events, keep = [], {}
        for row in rows:
            if row[7] != 1:
                continue
            key = (int(row[0]), int(row[2]))
            ev = cache.get(key) or make_decision_event(row, _SnapshotList(self, [e]))
            keep[key] = ev
            events.append(ev)
        self._joint_events[e] = keep
        self._n_pending[e] = len(events)
        return events

    # ---- scenario hooks (overridden by CitiBikeVectorEnv)
    def _encode_action(self, a) -> tuple:
        return encode_action(a)

    def _engine_step(self, acts, nact, mask, n_answered=None):
        self._host_ticks = None
        if n_answered is None:
            dec, met, done = self.engine.step(acts, nact, mask)
        else:
            dec, met, done = self.engine.step(acts, nact, mask, n_answered=n_answered)
        return dec.cpu().numpy(), met.cpu().numpy(), done.cpu().numpy(), None

    def _make_event(self, e: int, row, extra):
        return make_decision_event(row, _SnapshotList(self, [e]))

    def _node_counts(self) -> Dict[str, int]:
        t = self.engine.topo
        return {"ports": t.n_ports, "vessels": t.n_vessels, "matrices": 1}

    def _frame_index_lists(self) -> List[List[int]]:
        eng = self.engine
        fi = self._ring_fi_rows()
        S = fi.shape[1]
        ticks = eng.ticks.cpu().numpy()
        res = []
        for e in range(self._n):
            cur = (int(ticks[e]) - eng.start_tick) // eng.snapshot_resolution
            held = [int(x) for i, x in enumerate(fi[e]) if x >= 0 and not (self._paused[e] and i == cur % S)]
            if self._paused[e]:
                held.append(cur)     # the pre-decision snapshot (core.py:345) is the live frame
            res.append(sorted(held))
        return res

    def _agent_idx_list(self) -> List[int]:
        return list(range(self.engine.topo.n_ports))

    def _configs(self) -> dict:
        return self.engine.topo.raw_config

    def _name(self) -> str:
        return f"cim:{self.engine.topo.name}"

    def _summary(self) -> dict:
        t = self.engine.topo
        return {"node_mapping": {"ports": t.port_mapping, "vessels": t.vessel_mapping},
                "node_detail": {"ports": {"number": t.n_ports}, "vessels": {"number": t.n_vessels},
                                "matrices": {"number": 1}},
                "event_payload": {}}

    def _ring_fi_rows(self) -> np.ndarray:
        """int32 [n_envs, ring_slots]"""
        return self.engine.ring_fi.cpu().numpy()

    def _query(self, node: str, ticks: list, nodes: list, attrs: list) -> List[np.ndarray]:
        eng = self.engine
        n_nodes = self._node_counts()[node]
        nodes = list(range(n_nodes)) if not nodes else [int(x) for x in nodes]
        for x in nodes:
            if not -n_nodes <= x < n_nodes:
                raise IndexError(f"node index {x} out of range for {node}")
        nodes = [x % n_nodes for x in nodes]
        if ticks:
            t = np.tile(np.asarray(ticks, np.int32), (self._n, 1))
            counts = [len(ticks)] * self._n
        else:  # all stored frames of each env (np_backend.pyx:530-531) — ragged across envs
            lists = self._frame_index_lists()
            counts = [len(x) for x in lists]
            width = max(counts + [1])
            t = np.full((self._n, width), -1, np.int32)
            for e, x in enumerate(lists):
                t[e, :len(x)] = x
        out = eng.query(node, t, np.asarray(nodes, np.int32), attrs).cpu().numpy()
        return [out[e, :counts[e]].reshape(-1) for e in range(self._n)]


class GpuEnvView:
    """``AbsEnv``-shaped handle on one env of a ``GpuVectorEnv`` (core.py:92-260)."""

    def __init__(self, owner: GpuVectorEnv, index: int):
        self._o, self._i = owner, index
        self._snapshots = _SnapshotList(owner, [index])

    def step(self, action=None):
        return self._o._step_envs([self._i], {self._i: action})[0]

    def reset(self, keep_seed: bool = False):
        self._o.reset(keep_seed, [self._i])

    def set_seed(self, seed: int):
        assert seed is not None and isinstance(seed, int)
        self._o.set_seed(seed, [self._i])

    @property
    def tick(self) -> int:
        return int(self._o._ticks_host()[self._i])

    @property
    def frame_index(self) -> int:
        e = self._o.engine
        return (self.tick - e.start_tick) // e.snapshot_resolution

    @property
    def snapshot_list(self):
        return self._snapshots

    @property
    def agent_idx_list(self) -> List[int]:
        return self._o._agent_idx_list()

    @property
    def metrics(self) -> dict:
        m = self._o._last_met[self._i]
        return {k: int(m[i]) for i, k in enumerate(self._o.METRIC_KEYS)}

    @property
    def configs(self) -> dict:
        return self._o._configs()

    @property
    def name(self) -> str:
        return self._o._name()

    @property
    def summary(self) -> dict:
        return self._o._summary()

    def dump(self) -> None:
        """core.py:135-141: "Dump environment for restore.  NOTE: Not implemented." — a no-op there, a no-op here."""
        return None

    def get_finished_events(self):
        return []

    def get_pending_events(self, tick):
        return []

    @property
    def business_engine(self) -> "_BusinessEngineView":
        """abs_core.py:71.  The reference hands out the business engine object; callers read its frame / snapshots / metrics /
        agent list / node mapping (abs_business_engine.py:70-200).  Here that is a read-only view over this env of the batch."""
        return _BusinessEngineView(self)

    def get_ticks_frame_index_mapping(self) -> dict:
        """abs_core.py:174-195: tick -> frame index for the frames currently held in the snapshot list."""
        e = self._o.engine
        held = set(self.snapshot_list.get_frame_index_list())
        res, t0 = {}, e.start_tick
        for t in range(t0, self.tick + 1):
            fi = (t - t0) // e.snapshot_resolution
            if fi in held:
                res[t] = fi
        return res


class _BusinessEngineView:
    """``AbsBusinessEngine``-shaped read-only view (abs_business_engine.py:70-200) of one env: ``frame_index(tick)``,
    ``snapshots``, ``get_metrics()``, ``get_agent_idx_list()``, ``get_node_mapping()``, ``configs``, ``name``,
    ``scenario_name``.  ``frame`` is the snapshot list's owner in the reference; code that only reads it through
    ``frame.snapshots`` keeps working (``frame.snapshots is snapshots``)."""

    def __init__(self, env: GpuEnvView):
        self._env = env

    @property
    def frame(self):
        return self

    @property
    def snapshots(self):
        return self._env.snapshot_list

    @property
    def configs(self) -> dict:
        return self._env.configs

    @property
    def name(self) -> str:
        return self._env.name

    @property
    def scenario_name(self) -> str:
        return self._env.name.split(":")[0]

    def frame_index(self, tick: int) -> int:
        e = self._env._o.engine
        return (tick - e.start_tick) // e.snapshot_resolution

    def calc_max_snapshots(self) -> int:
        e = self._env._o.engine
        return int(e.layout.ring_slots)

    def get_metrics(self) -> dict:
        return self._env.metrics

    def get_agent_idx_list(self) -> List[int]:
        return self._env.agent_idx_list

    def get_node_mapping(self) -> dict:
        return self._env.summary["node_mapping"]

    def get_event_payload_detail(self) -> dict:
        return self._env.summary.get("event_payload", {})
