"""Batched citi_bike rollout engine: thin Python plumbing over the C ABI (include/maro_amd_citi_bike.h).

PyTorch owns the device memory and the stream; every state transition runs in the HIP kernels of
maro_amd/csrc/cb_engine.hip.  Tensor conventions are those of the C ABI:
    actions   int32 [n_envs, A, 3] = (from_station_idx, to_station_idx, number)
    decisions int32 [n_envs, 8]    = (tick, station_idx, 0=Supply | 1=Demand, frame_index, n_scope, valid, 0, 0)
    scope     int32 [n_envs, K, 2] = ordered (station, max) pairs of the decision's action_scope, the deciding station last
    metrics   int64 [n_envs, 3]    = (trip_requirements, bike_shortage, operation_number)
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence, Union

import numpy as np
import torch

from .. import _lib
from .abi import (HDR_STATUS, HDR_TICK, HDR_WORDS, NODE_TYPE, MrxCbConfig, MrxCbLayout, draw_transfer_times,
                  topology_struct)
from .data import CitiBikeData, load_topology


class CitiBikeBatchEngine:
    """N independent citi_bike environments resident on one MI355X (`decision_mode` 0 Sequential: `step`; 1 Joint /
    2 JointWithSequentialAction: `step_joint`, decisions [n, S, 8], scope [n, S, scope_cap, 2])."""

    def __init__(self, topology: Union[str, CitiBikeData], n_envs: int, start_tick: int = 0, durations: int = 1440,
                 snapshot_resolution: int = 1, max_snapshots: Optional[int] = None, max_actions: int = 1,
                 device: Union[str, torch.device] = "cuda:0", seeds: Optional[Sequence[int]] = None,
                 delivery_capacity: int = 0, transfer_times_cap: int = 0, specialize: Union[bool, str, None] = None,
                 decision_mode: int = 0):
        """specialize: as CimBatchEngine — True = reset / step kernels compiled for this exact plan (maro_amd/cim/specialize.py,
        a few seconds of hipcc the first time, cached in-tree), "cached" = only if already cached, False = generic kernels,
        None = $MARO_AMD_SPECIALIZE."""
        if specialize is None:
            specialize = {"1": True, "cached": "cached"}.get(os.environ.get("MARO_AMD_SPECIALIZE", "0"), False)
        self._L = _lib.load()  # raises if the HIP extension is not built
        if not torch.cuda.is_available():
            raise RuntimeError("maro_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU path")
        self.data = topology if isinstance(topology, CitiBikeData) else load_topology(topology)
        self.device = torch.device(device)
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.n_envs, self.max_actions = int(n_envs), int(max_actions)
        self.start_tick, self.durations = int(start_tick), int(durations)
        self.max_tick = self.start_tick + self.durations
        self.snapshot_resolution = int(snapshot_resolution)
        if not delivery_capacity:
            # every decision tick can put one DeliverBike per station in flight for up to ~mean + 6 std ticks
            d = self.data
            delivery_capacity = d.n_stations * (int((d.time_mean + 6 * d.time_std) / max(d.resolution, 1)) + 2) + 4
        self._ts, self._keep_topo = topology_struct(self.data)
        self._cfg = MrxCbConfig(self.n_envs, dev_index, self.start_tick, self.durations, self.snapshot_resolution,
                                int(max_snapshots or 0), self.max_actions, int(delivery_capacity), int(transfer_times_cap),
                                int(decision_mode))
        self.decision_mode = int(decision_mode)
        nbytes = self._L.mrx_cb_workspace_bytes(ctypes.byref(self._ts), ctypes.byref(self._cfg))
        _lib.check(nbytes, "mrx_cb_workspace_bytes")
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            torch.cuda.synchronize()
            _lib.check(self._L.mrx_cb_create(ctypes.byref(self._ts), ctypes.byref(self._cfg), self.workspace.data_ptr(), nbytes,
                                             ctypes.byref(h)), "mrx_cb_create")
        self._h = h
        self.specialized = False
        self._specialize = specialize
        self._forced_wave = False
        self._manual_lanes = bool(os.environ.get("MRX_CB_LANES"))   # (the C side reads it at create: an experiment's fixed split)
        self._load_specialized()
        self.layout = MrxCbLayout()
        _lib.check(self._L.mrx_cb_get_layout(self._h, ctypes.byref(self.layout)), "mrx_cb_get_layout")
        lay = self.layout
        st = lay.env_stride
        # zero-copy views of engine state, indexed [word, env] whatever the layout in memory (struct-of-arrays [word][env stride],
        # or — lay.env_major: plans that step one env per wave — [env][words], seen through a transpose)
        if lay.env_major:
            self.hdr = self._view(lay.off_hdr, (st, HDR_WORDS))[:self.n_envs].t()
            self.live = self._view(lay.off_live, (st, lay.frame_words))[:self.n_envs].t()
            self.ring = self._view(lay.off_ring, (st, lay.ring_slots, lay.frame_words + 1))[:self.n_envs].permute(1, 2, 0)
            self.ring_fi = self._view(lay.off_ring_fi, (st, lay.ring_slots))[:self.n_envs].t()
        else:
            self.hdr = self._view(lay.off_hdr, (HDR_WORDS, st))[:, :self.n_envs]
            self.live = self._view(lay.off_live, (lay.frame_words, st))[:, :self.n_envs]
            self.ring = self._view(lay.off_ring, (lay.ring_slots, lay.frame_words + 1, st))[:, :, :self.n_envs]
            self.ring_fi = self._view(lay.off_ring_fi, (lay.ring_slots, st))[:, :self.n_envs]
        self.ticks, self.status = self.hdr[HDR_TICK], self.hdr[HDR_STATUS]
        rows = (self.data.n_stations,) if self.decision_mode else ()
        self.decisions = torch.zeros((self.n_envs,) + rows + (8,), dtype=torch.int32, device=self.device)
        self.scope = torch.full((self.n_envs,) + rows + (lay.scope_cap, 2), -1, dtype=torch.int32, device=self.device)
        self.metrics = torch.zeros((self.n_envs, 3), dtype=torch.int64, device=self.device)
        self.done = torch.zeros((self.n_envs,), dtype=torch.uint8, device=self.device)
        if seeds is not None:
            self.reset(seeds=seeds)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._L.mrx_cb_destroy(h)
            self._h = None

    def _view(self, off: int, shape) -> torch.Tensor:
        n = int(np.prod(shape)) * 4
        return self.workspace[off:off + n].view(torch.int32).view(*shape)

    def use_stream(self, stream: Optional[torch.cuda.Stream]) -> None:
        """Bind every later call of this engine to `stream` (None: back to torch's current stream at call time) — the same stream
        discipline as ``CimBatchEngine.use_stream``: kernels launch on it, host inputs are converted on it, device tensors passed as
        they are must have been produced on it (or be complete), outputs are read on it or after ``stream.synchronize()``.  A rollout
        loop that drives several env groups on their own streams overlaps their (latency-bound) batch steps this way."""
        self._bound_stream = stream
        self._bound_handle = None if stream is None else stream.cuda_stream

    def _stream(self) -> int:
        h = getattr(self, "_bound_handle", None)
        return torch.cuda.current_stream(self.device).cuda_stream if h is None else h

    def _dev(self, x, dtype) -> Optional[torch.Tensor]:
        if x is None:
            return None
        if isinstance(x, torch.Tensor) and x.dtype == dtype and x.device == self.device and x.is_contiguous():
            return x   # the usual case in a rollout loop: no conversion, no extra launch
        if not isinstance(x, torch.Tensor):
            x = torch.as_tensor(np.asarray(x), dtype=dtype)
        bound = getattr(self, "_bound_stream", None)
        if bound is None:
            return x.to(device=self.device, dtype=dtype).contiguous()
        if x.is_cuda:
            bound.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(bound):   # the copy / cast is ordered before the kernel on the bound stream
            return x.to(device=self.device, dtype=dtype).contiguous()

    @staticmethod
    def _p(t: Optional[torch.Tensor]):
        return None if t is None else t.data_ptr()

    # ------------------------------------------------------------------ C ABI calls
    def reset(self, seeds: Optional[Sequence[int]] = None, transfer_times=None, mask=None) -> None:
        """`seeds`: one numpy seed per env — env e replays what `np.random.seed(seeds[e])` gives the reference's
        transfer times; or pass the `transfer_times` [n_envs, k] directly; neither = keep the streams in place."""
        if seeds is not None:
            transfer_times = draw_transfer_times(self.data, seeds, self.layout.transfer_times_cap)
        tt = self._dev(transfer_times, torch.int32)
        mk = self._dev(mask, torch.uint8)
        if tt is not None:
            assert tt.dim() == 2 and tt.shape[0] == self.n_envs, "transfer_times must be [n_envs, k]"
        _lib.check(self._L.mrx_cb_reset(self._h, self._p(tt), 0 if tt is None else int(tt.shape[1]), self._p(mk), self._stream()),
                   "mrx_cb_reset")
        self._keep = (tt, mk)

    def _load_specialized(self, runtime_shift: Optional[bool] = None) -> None:
        """(Re)load the step kernels compiled for this plan.  The plan text carries the envs-per-wave shift of the automatic split
        (MRXC_lsh_plan: folded into the kernel's LDS addresses); with a split chosen by hand — or the wave-cooperative kernels forced
        on, which only the other build contains — the build that takes the shift as a kernel argument is loaded instead
        (MRXC_lsh_plan -1).  Raises what the load raises (the previously loaded kernels stay in place); in "cached" mode a build that
        is not in the cache leaves everything as it was, `specialized` included."""
        if not self._specialize:
            self.specialized = False
            return
        import re

        from ..cim import specialize as spec
        if runtime_shift is None:
            runtime_shift = self._manual_lanes or self._forced_wave
        defines = spec.plan_defines(self._ts, self._cfg, "citi_bike")
        if runtime_shift:
            defines = re.sub(r"(#define MRXC_lsh_plan) -?\d+", r"\1 -1", defines)
        try:
            spec.load_into(self, defines, build=self._specialize != "cached", scenario="citi_bike")
            self.specialized = True
        except KeyError:
            pass   # "cached" and not in the cache: whatever was loaded before (generic kernels at creation) stays

    def set_lanes_per_wave(self, lanes: int = 0) -> None:
        """Envs per 64-lane wave of the step kernel (1, 2, ..., 64; 0 = the automatic choice): few envs per wave = less
        control-flow divergence, more waves.  Results do not depend on it (include/maro_amd_citi_bike.h).  A plan-specialised engine
        reloads its step kernels when the choice switches between automatic and by hand (see _load_specialized): the matching build
        is loaded FIRST, so a refused load leaves the split and the kernels as they were; a refused split reloads the old build."""
        manual = int(lanes) != 0
        switch = bool(self._specialize) and (manual or self._forced_wave) != (self._manual_lanes or self._forced_wave)
        if switch:
            self._load_specialized(manual or self._forced_wave)
        try:
            _lib.check(self._L.mrx_cb_set_lanes_per_wave(self._h, int(lanes)), "mrx_cb_set_lanes_per_wave")
        except Exception:
            if switch:
                self._load_specialized()
            raise
        self._manual_lanes = manual

    def set_wave_decisions(self, mode: int = 0) -> bool:
        """mrx_cb_set_wave_decisions: env-steps that stay inside their tick on one wave per env (action scope ranked across the
        lanes).  0 = automatic (on from 96 stations), 1 = on, -1 = off; returns whether it is in effect.  Results unchanged.
        Forcing it ON for a plan-specialised engine whose plan fits LDS (compiled-in envs-per-wave shift) loads the runtime-shift
        build first: only that one contains the wave kernels (cb_step_kernels.h)."""
        forced = int(mode) == 1
        switch = bool(self._specialize) and (self._manual_lanes or forced) != (self._manual_lanes or self._forced_wave)
        if switch:
            self._load_specialized(self._manual_lanes or forced)
        try:
            on = bool(_lib.check(self._L.mrx_cb_set_wave_decisions(self._h, int(mode)), "mrx_cb_set_wave_decisions"))
        except Exception:
            if switch:
                self._load_specialized()
            raise
        self._forced_wave = forced
        return on

    def set_replay_overlap(self, on: bool = True) -> None:
        """mrx_cb_set_replay_overlap: the in-tick kernel and the replay kernel of a batch step side by side (plan-specialised wave
        kernels) or one after the other (the default).  Results unchanged."""
        _lib.check(self._L.mrx_cb_set_replay_overlap(self._h, int(bool(on))), "mrx_cb_set_replay_overlap")

    def set_replay_period(self, n: int = 1, phase: int = 0) -> None:
        """mrx_cb_set_replay_period: the replay kernel of the wave-stepped path on every n-th `step()` only (use a step budget n times
        as large: the same replaying per call at 1 / n of the kernel's fixed cost).  In between, envs that leave their tick keep the
        answer they were given and report `decisions[e, 5] == 0`, as under a step budget.  Trajectories unchanged.  `phase` (0 .. n - 1)
        shifts which calls those are: env groups on streams of their own take different phases."""
        _lib.check(self._L.mrx_cb_set_replay_period(self._h, int(n), int(phase)), "mrx_cb_set_replay_period")

    def set_step_budget(self, max_records: int = 0) -> None:
        """Bounded steps: an env replays at most ~`max_records` events per `step()` call; envs that have not reached their
        next decision report `decisions[e, 5] == 0` (and `done[e] == 0`) and continue in the next call.  0 = off."""
        _lib.check(self._L.mrx_cb_set_step_budget(self._h, int(max_records)), "mrx_cb_set_step_budget")

    def set_observation(self, station_attrs: Sequence[str] = ()) -> Optional[torch.Tensor]:
        """Fuse an agent's per-decision snapshot slice into step() (mrx_cb_set_observation): returns float64 [n_envs, rows, len(attrs)],
        rewritten by every step() for each env's new decision — the same values as a `query("stations", decisions[:, 3:4], nodes, attrs)`,
        without the extra launch.  Plans stepped one env per lane: rows = every station.  Plans stepped by the wave-cooperative kernels
        (`layout.env_major`, plan-specialised): rows = the `scope_cap` stations of the decision's action scope, `scope[:, :, 0]`
        (-1 padding: zeros).  () switches it off."""
        ids = self.attr_ids("stations", station_attrs)
        rows = _lib.check(self._L.mrx_cb_observation_rows(self._h), "mrx_cb_observation_rows")   # read-only: the wave mode the caller chose stays
        self.obs = torch.zeros((self.n_envs, rows, len(ids)), dtype=torch.float64, device=self.device) if ids else None
        arr = (ctypes.c_int32 * max(len(ids), 1))(*ids)
        _lib.check(self._L.mrx_cb_set_observation(self._h, arr, len(ids), self._p(self.obs)), "mrx_cb_set_observation")
        return self.obs

    def step(self, actions=None, n_actions=None, mask=None):
        a = self._dev(actions, torch.int32)
        na = self._dev(n_actions, torch.int32)
        mk = self._dev(mask, torch.uint8)
        if a is not None:
            assert a.numel() == self.n_envs * self.max_actions * 3, "actions must be [n_envs, max_actions, 3]"
            if na is None:
                na = torch.full((self.n_envs,), self.max_actions, dtype=torch.int32, device=self.device)
        _lib.check(self._L.mrx_cb_step(self._h, self._p(a), self._p(na), self._p(mk), self.decisions.data_ptr(), self.scope.data_ptr(),
                                       self.metrics.data_ptr(), self.done.data_ptr(), self._stream()), "mrx_cb_step")
        self._keep = (a, na, mk)
        return self.decisions, self.scope, self.metrics, self.done

    def step_joint(self, actions=None, n_actions=None, n_answered=None, mask=None):
        """Joint decision modes: `actions` int32 [n, S, A, 3] (row i = the action list of the i-th reported event),
        `n_actions` int32 [n, S], `n_answered` int32 [n] (include/maro_amd_citi_bike.h::mrx_cb_step_joint)."""
        a = self._dev(actions, torch.int32)
        na = self._dev(n_actions, torch.int32)
        nans = self._dev(n_answered, torch.int32)
        mk = self._dev(mask, torch.uint8)
        S = self.data.n_stations
        if a is not None:
            assert a.numel() == self.n_envs * S * self.max_actions * 3, "actions must be [n_envs, S, max_actions, 3]"
            assert na is not None and na.numel() == self.n_envs * S and nans is not None and nans.numel() == self.n_envs
        _lib.check(self._L.mrx_cb_step_joint(self._h, self._p(a), self._p(na), self._p(nans), self._p(mk), self.decisions.data_ptr(),
                                             self.scope.data_ptr(), self.metrics.data_ptr(), self.done.data_ptr(), self._stream()), "mrx_cb_step_joint")
        self._keep_step = (a, na, nans, mk)
        return self.decisions, self.scope, self.metrics, self.done

    def random_policy(self, step: int, actions: torch.Tensor, n_actions: torch.Tensor, counter: Optional[torch.Tensor] = None) -> None:
        _lib.check(self._L.mrx_cb_random_policy(self._h, self.decisions.data_ptr(), self.scope.data_ptr(), int(step), actions.data_ptr(),
                                                n_actions.data_ptr(), self._p(counter), self._stream()), "mrx_cb_random_policy")

    def attr_ids(self, node: str, attrs: Sequence[str]):
        ids = []
        for a in attrs:
            i = self._L.mrx_cb_attr_id(NODE_TYPE[node], a.encode())
            if i < 0:
                raise KeyError(f"unknown attribute {a!r} of node {node!r}")
            ids.append(i)
        return ids

    def row_slots(self, node: str, attr_ids: Sequence[int]) -> int:
        return sum(self._L.mrx_cb_attr_slots(self._h, NODE_TYPE[node], int(a)) for a in attr_ids)

    def query(self, node: str, ticks, nodes, attrs: Sequence[str], out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """float64 [n_envs, nt, nn, sum(slots)]; `ticks` are frame indices [nt] or [n_envs, nt]; `nodes` [nn] or [n_envs, nn]."""
        ids = self.attr_ids(node, attrs)

        def rows(x):
            if isinstance(x, torch.Tensor) and x.dim() == 2 and x.dtype == torch.int32 and x.is_cuda and x.stride(1) == 1 and x.stride(0) > 0:
                return x, int(x.stride(0))
            x = self._dev(x, torch.int32)
            return x, (int(x.shape[1]) if x.dim() == 2 else 0)

        t, per_env = rows(ticks)
        n, nodes_per_env = rows(nodes)
        nt, nn = int(t.shape[-1]), int(n.shape[-1])
        slots = self.row_slots(node, ids)
        if out is None:
            out = torch.empty((self.n_envs, nt, nn, slots), dtype=torch.float64, device=self.device)
        ida = (ctypes.c_int32 * len(ids))(*ids)
        _lib.check(self._L.mrx_cb_query(self._h, NODE_TYPE[node], t.data_ptr(), nt, per_env, n.data_ptr(), nn, nodes_per_env, ida, len(ids),
                                        out.data_ptr(), self._stream()), "mrx_cb_query")
        self._keep_q = (t, n)
        return out
