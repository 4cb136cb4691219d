"""Batched CIM rollout engine: thin Python plumbing over the C ABI (include/maro_amd.h).

PyTorch owns the device memory (one uint8 workspace tensor + caller-visible I/O tensors) and the
stream; every state transition runs in the hand-written HIP kernels of maro_amd/csrc/.
Tensor conventions are those of the C ABI:
    actions   int32 [n_envs, A, 4] = (vessel_idx, port_idx, quantity, 0=LOAD | 1=DISCHARGE)
    decisions int32 [n_envs, 8]    = (tick, port_idx, vessel_idx, scope.load, scope.discharge,
                                      early_discharge, frame_index, valid)
    metrics   int64 [n_envs, 3]    = (order_requirements, container_shortage, operation_number)
"""
from __future__ import annotations

import contextlib
import ctypes
import os
from typing import Optional, Sequence, Union

import numpy as np
import torch

from .. import _lib
from .topology import CimTopology, load_topology

PORT_ATTRS = ["capacity", "empty", "full", "on_shipper", "on_consignee", "shortage", "acc_shortage", "booking",
              "acc_booking", "fulfillment", "acc_fulfillment", "transfer_cost"]
VESSEL_ATTRS = ["capacity", "empty", "full", "remaining_space", "early_discharge", "is_parking", "loc_port_idx",
                "route_idx", "last_loc_idx", "next_loc_idx", "past_stop_list", "past_stop_tick_list",
                "future_stop_list", "future_stop_tick_list"]
MATRIX_ATTRS = ["full_on_ports", "full_on_vessels", "vessel_plans"]
NODE_ATTRS = {"ports": PORT_ATTRS, "vessels": VESSEL_ATTRS, "matrices": MATRIX_ATTRS}
NODE_TYPE = {"ports": 0, "vessels": 1, "matrices": 2}

SEED_KEEP = -1      # reset(keep_seed=True)
SEED_REDRAW = -2    # reset(keep_seed=False): new seed = route_init.randint(0, 4095)


class CimBatchEngine:
    """N independent CIM environments (Sequential decision mode) resident on one MI355X."""

    def __init__(self, topology: Union[str, CimTopology], n_envs: int, start_tick: int = 0, durations: int = 100,
                 snapshot_resolution: int = 1, max_snapshots: Optional[int] = None, max_actions: int = 1,
                 device: Union[str, torch.device] = "cuda:0", seeds: Optional[Sequence[int]] = None, order_table: int = 0,
                 decision_mode: int = 0, specialize: Union[bool, str, None] = None, step_mode: int = 0):
        """step_mode: how step() launches its work (mrx_cim_set_step_mode): 0 = default (the sorted launch), 1 = unsorted,
        2 = sorted, 3 = persistent pipelined kernel (needs specialised kernels and the order table).  Pure scheduling:
        results are identical.
        specialize: True = step with kernels compiled for this exact plan (maro_amd/cim/specialize.py: ~15 s of hipcc
        the first time a (topology, config) is seen, cached in-tree; +12 % env-steps/s); "cached" = use them only if the
        code object is already in the cache; False = the generic kernels; None = $MARO_AMD_SPECIALIZE ("1" / "cached" / "0"),
        default generic."""
        if specialize is None:
            specialize = {"1": True, "cached": "cached"}.get(os.environ.get("MARO_AMD_SPECIALIZE", "0"), False)
        self._L = _lib.load()  # raises if the HIP extension is not built
        if not torch.cuda.is_available():
            raise RuntimeError("maro_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU path")
        self.topo = topology if isinstance(topology, CimTopology) else load_topology(topology)
        self.device = torch.device(device)
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.n_envs, self.max_actions = int(n_envs), int(max_actions)
        self.start_tick, self.durations = int(start_tick), int(durations)
        self.max_tick = self.start_tick + self.durations
        self.snapshot_resolution = int(snapshot_resolution)
        self._cs = self.topo.c_struct()
        self._cfg = _lib.MrxCimConfig(self.n_envs, dev_index, self.start_tick, self.durations,
                                      self.snapshot_resolution, int(max_snapshots or 0), self.max_actions, 0, int(decision_mode),
                                      int(order_table))
        self.decision_mode = int(decision_mode)
        nbytes = self._L.mrx_cim_workspace_bytes(ctypes.byref(self._cs), ctypes.byref(self._cfg))
        _lib.check(nbytes, "mrx_cim_workspace_bytes")
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            torch.cuda.synchronize()
            _lib.check(self._L.mrx_cim_create(ctypes.byref(self._cs), ctypes.byref(self._cfg),
                                              self.workspace.data_ptr(), nbytes, ctypes.byref(h)), "mrx_cim_create")
        self._h = h
        self._bound_stream, self._bound_handle = None, None
        self.specialized = False
        self._specialize, self._obs_ids = specialize, ((), ())
        self._step_mode = int(step_mode)
        self._load_specialized()
        self.layout = _lib.MrxCimLayout()
        _lib.check(self._L.mrx_cim_get_layout(self._h, ctypes.byref(self.layout)), "mrx_cim_get_layout")
        lay = self.layout
        # zero-copy views of engine state
        self.live = self._view(lay.off_live, torch.int32, (self.n_envs, lay.frame_words))
        self.ring = self._view(lay.off_ring, torch.int32, (self.n_envs, lay.ring_slots, lay.frame_words))
        self.ring_fi = self._view(lay.off_ring_fi, torch.int32, (self.n_envs, lay.ring_slots))
        self.status = self._view(lay.off_status, torch.int32, (self.n_envs,))
        self.ticks = self._view(lay.off_tick, torch.int32, (self.n_envs,))
        self.seeds = self._view(lay.off_seed, torch.int64, (self.n_envs,))
        # persistent outputs
        dshape = (self.n_envs, 8) if self.decision_mode == 0 else (self.n_envs, lay.n_vessels, 8)  # Joint: a row per vessel
        self.decisions = torch.zeros(dshape, dtype=torch.int32, device=self.device)
        self.metrics = torch.zeros((self.n_envs, 3), dtype=torch.int64, device=self.device)
        self.done = torch.zeros((self.n_envs,), dtype=torch.uint8, device=self.device)
        if seeds is not None:
            self.reset(seeds)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._L.mrx_cim_destroy(h)
            self._h = None

    # ------------------------------------------------------------------ helpers
    def _view(self, off: int, dtype: torch.dtype, shape) -> torch.Tensor:
        n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        return self.workspace[off:off + n].view(dtype).view(*shape)

    def _load_specialized(self) -> None:
        """(Re)load the step kernels compiled for this plan and the current fused-observation configuration."""
        self.specialized = False
        if self._specialize:
            from . import specialize as spec
            try:
                spec.load_into(self, spec.plan_defines(self._cs, self._cfg, obs=self._obs_ids), build=self._specialize != "cached")
                self.specialized = True
            except KeyError:
                pass   # "cached" and not in the cache: generic kernels
        self.set_step_mode(self._step_mode)

    def set_step_mode(self, mode: int) -> int:
        """Select the launch form of step() (see __init__); returns — and keeps in `step_mode` — the form in effect."""
        self._step_mode = int(mode)
        self.step_mode = _lib.check(self._L.mrx_cim_set_step_mode(self._h, self._step_mode), "mrx_cim_set_step_mode")
        return self.step_mode

    def use_stream(self, stream: Optional[torch.cuda.Stream]) -> None:
        """Bind every later call of this engine to `stream` (None: back to torch's current stream at call time).  A rollout
        loop that drives several engines on their own streams saves the per-call stream lookup / context switch.

        Stream discipline while a stream is bound: kernels launch on it; inputs that need a conversion / host-to-device copy
        are converted ON it (so the copy is ordered before the kernel, and the allocator ties the temporary to that stream);
        device tensors passed as they are must have been produced on it (or be complete).  Outputs (`decisions`, `metrics`,
        `done`, observation buffers, query results) must be read on the bound stream (`with torch.cuda.stream(s): ...`) or
        after `s.synchronize()` — a side stream does not synchronise with torch's default stream."""
        self._bound_stream = stream
        self._bound_handle = None if stream is None else stream.cuda_stream

    def _stream(self) -> int:
        h = self._bound_handle
        return torch.cuda.current_stream(self.device).cuda_stream if h is None else h

    def _dev(self, x, dtype) -> Optional[torch.Tensor]:
        if x is None:
            return None
        if isinstance(x, torch.Tensor) and x.dtype == dtype and x.device == self.device and x.is_contiguous():
            return x   # the usual case in a rollout loop: no conversion, no extra launch
        if not isinstance(x, torch.Tensor):
            x = torch.as_tensor(np.asarray(x), dtype=dtype)
        if self._bound_stream is None:
            return x.to(device=self.device, dtype=dtype).contiguous()
        # a bound (side) stream does not wait for torch's current stream: the copy / cast must be enqueued on the bound stream
        # itself, or the kernel could read the tensor before a pageable host-to-device copy has landed.  A device tensor
        # produced on the current stream is waited for first.
        if x.is_cuda:
            self._bound_stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self._bound_stream):
            return x.to(device=self.device, dtype=dtype).contiguous()

    @staticmethod
    def _p(t: Optional[torch.Tensor]):
        return None if t is None else t.data_ptr()

    # ------------------------------------------------------------------ C ABI calls
    def reset(self, seed_cmd=None, mask=None) -> None:
        """seed_cmd[e]: >=0 explicit seed (set_seed + reset(keep_seed=True)); -1 keep; -2 redraw."""
        sc = self._dev(seed_cmd, torch.int64)
        mk = self._dev(mask, torch.uint8)
        _lib.check(self._L.mrx_cim_reset(self._h, self._p(sc), self._p(mk), self._stream()), "mrx_cim_reset")
        self._keep = (sc, mk)

    def step(self, actions=None, n_actions=None, mask=None, n_answered=None):
        """Sequential mode: decisions [n, 8].  Joint modes (decision_mode 1 / 2): decisions [n, V, 8], one row per pending
        event; `actions` is the flat list of the answered events' actions and `n_answered` (mode 2) how many events
        they answer (None = all)."""
        a = self._dev(actions, torch.int32)
        na = self._dev(n_actions, torch.int32)
        mk = self._dev(mask, torch.uint8)
        if a is not None:
            assert a.numel() == self.n_envs * self.max_actions * 4, "actions must be [n_envs, max_actions, 4]"
            if na is None:
                na = torch.full((self.n_envs,), self.max_actions, dtype=torch.int32, device=self.device)
        if self.decision_mode == 0:
            _lib.check(self._L.mrx_cim_step(self._h, self._p(a), self._p(na), self._p(mk), self.decisions.data_ptr(),
                                            self.metrics.data_ptr(), self.done.data_ptr(), self._stream()), "mrx_cim_step")
            self._keep = (a, na, mk)
        else:
            nans = self._dev(n_answered, torch.int32)
            _lib.check(self._L.mrx_cim_step_joint(self._h, self._p(a), self._p(na), self._p(nans), self._p(mk), self.decisions.data_ptr(),
                                                  self.metrics.data_ptr(), self.done.data_ptr(), self._stream()), "mrx_cim_step_joint")
            self._keep = (a, na, mk, nans)
        return self.decisions, self.metrics, self.done

    def clear_status_bits(self, envs, bits: int) -> None:
        """status[e] &= ~bits for the listed envs (the object API acknowledges MRX_ENV_INVALID_ACTION this way)."""
        st = self._bound_stream
        with (torch.cuda.stream(st) if st is not None else contextlib.nullcontext()):
            idx = torch.as_tensor(list(envs), dtype=torch.int64, device=self.status.device)
            self.status[idx] = self.status[idx] & ~int(bits)

    def random_policy(self, step: int, actions: torch.Tensor, n_actions: torch.Tensor,
                      counter: Optional[torch.Tensor] = None) -> None:
        """Device-side random legal agent (hello-world policy): fills actions[:,0] / n_actions from the current
        decisions; adds the number of answered decisions to `counter` (uint64/int64 scalar tensor)."""
        _lib.check(self._L.mrx_cim_random_policy(self._h, self.decisions.data_ptr(), int(step), actions.data_ptr(),
                                                 n_actions.data_ptr(), self._p(counter), self._stream()),
                   "mrx_cim_random_policy")

    def set_device_agent(self, actions: Optional[torch.Tensor] = None, n_actions: Optional[torch.Tensor] = None,
                         counts: Optional[torch.Tensor] = None, next_key: int = -1) -> None:
        """mrx_cim_set_device_agent: the random legal agent answered INSIDE step() — every step writes the action
        ``random_policy(next_key, ...)`` would write for the decision it has just raised into `actions` / `n_actions` (the tensors
        the next ``step(actions, n_actions)`` is given) and adds 1 to `counts[e]` (int32 [n_envs]).  `next_key` >= 0 counts up
        with every step() call that follows (set it again after a reset); < 0: keyed on the decision's (tick, vessel).
        ``set_device_agent()`` switches it off.  The tensors must outlive the setting."""
        on = actions is not None
        if on:
            assert actions.dtype == torch.int32 and n_actions.dtype == torch.int32 and actions.numel() == self.n_envs * self.max_actions * 4
            assert counts is None or (counts.dtype == torch.int32 and counts.numel() == self.n_envs)
        _lib.check(self._L.mrx_cim_set_device_agent(self._h, 1 if on else 0, self._p(actions), self._p(n_actions), self._p(counts), int(next_key)),
                   "mrx_cim_set_device_agent")
        self._agent_keep = (actions, n_actions, counts)

    def set_observation(self, port_attrs: Sequence[str] = (), vessel_attrs: Sequence[str] = ()):
        """Fuse an agent's per-decision snapshot slices into step(): returns (obs_ports float64 [n, P, len(port_attrs)],
        obs_vessel float64 [n, len(vessel_attrs)]), rewritten by every step() for the envs that pause at a new decision —
        the values `query("ports", decisions[:, 6:7], all ports, port_attrs)` and
        `query("vessels", decisions[:, 6:7], decisions[:, 2:3], vessel_attrs)` would return, without the extra launches."""
        pa, va = self.attr_ids("ports", port_attrs), self.attr_ids("vessels", vessel_attrs)
        self.obs_ports = torch.zeros((self.n_envs, self.layout.n_ports, len(pa)), dtype=torch.float64, device=self.device)
        self.obs_vessel = torch.zeros((self.n_envs, len(va)), dtype=torch.float64, device=self.device)
        pa_c, va_c = (ctypes.c_int32 * max(len(pa), 1))(*pa), (ctypes.c_int32 * max(len(va), 1))(*va)
        _lib.check(self._L.mrx_cim_set_observation(self._h, pa_c, len(pa), va_c, len(va), self.obs_ports.data_ptr(),
                                                   self.obs_vessel.data_ptr()), "mrx_cim_set_observation")
        if (tuple(pa), tuple(va)) != tuple(map(tuple, self._obs_ids)):
            self._obs_ids = (tuple(pa), tuple(va))
            self._load_specialized()   # the observation's configuration is compiled into the specialised kernels
        return self.obs_ports, self.obs_vessel

    def set_port_history(self, port_attrs: Sequence[str] = ()) -> Optional[torch.Tensor]:
        """Per-attribute retention (mrx_cim_set_port_history): returns int32 [n_envs, frames, len(port_attrs), n_ports], into
        which every snapshot of an env also writes these port attributes of its frame — the whole episode of a few
        attributes (e.g. fulfillment / shortage for the CIM example's delayed reward) next to a short snapshot ring.
        Rows are NOT cleared by reset(): zero them (`hist[envs] = 0`) when envs start a new episode.  () switches it off."""
        ids = self.attr_ids("ports", port_attrs)
        frames = -(-self.durations // self.snapshot_resolution)
        self.port_history = torch.zeros((self.n_envs, frames, len(ids), self.layout.n_ports), dtype=torch.int32, device=self.device) if ids else None
        arr = (ctypes.c_int32 * max(len(ids), 1))(*ids)
        _lib.check(self._L.mrx_cim_set_port_history(self._h, arr, len(ids), self._p(self.port_history), frames), "mrx_cim_set_port_history")
        return self.port_history

    def attr_ids(self, node: str, attrs: Sequence[str]):
        ids = []
        for a in attrs:
            i = self._L.mrx_cim_attr_id(NODE_TYPE[node], a.encode())
            if i < 0:
                raise KeyError(f"unknown attribute {a!r} of node {node!r}")
            ids.append(i)
        return ids

    def row_slots(self, node: str, attr_ids: Sequence[int]) -> int:
        return sum(self._L.mrx_cim_attr_slots(self._h, NODE_TYPE[node], int(a)) for a in attr_ids)

    def query(self, node: str, ticks, nodes, attrs: Sequence[str], out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """float64 [n_envs, nt, nn, sum(slots)]; `ticks` are frame indices, [nt] or [n_envs, nt]; `nodes` [nn] or
        per-env [n_envs, nn] (out-of-range node indices, e.g. the -1 padding of stop lists, give zeros)."""
        ids = self.attr_ids(node, attrs)
        if isinstance(ticks, torch.Tensor) and ticks.dim() == 2 and ticks.dtype == torch.int32 and ticks.is_cuda \
                and ticks.stride(1) == 1:
            t = ticks                      # strided per-env rows (e.g. decisions[:, 6:7]) are passed through without a copy
        else:
            t = self._dev(ticks, torch.int32)
        if isinstance(nodes, torch.Tensor) and nodes.dim() == 2 and nodes.dtype == torch.int32 and nodes.is_cuda \
                and nodes.stride(1) == 1:
            n = nodes
        else:
            n = self._dev(nodes, torch.int32)
        nodes_per_env = int(n.stride(0)) if n.dim() == 2 else 0
        if n.dim() == 2 and nodes_per_env == 0:
            n, nodes_per_env = n.contiguous(), int(n.shape[1])
        per_env = int(t.stride(0)) if t.dim() == 2 else 0
        if t.dim() == 2 and per_env == 0:
            t, per_env = t.contiguous(), int(t.shape[1])
        nt, nn = int(t.shape[-1]), int(n.shape[-1])
        slots = self.row_slots(node, ids)
        if out is None:
            if self._bound_stream is None:
                out = torch.empty((self.n_envs, nt, nn, slots), dtype=torch.float64, device=self.device)
            else:   # allocated on the stream that writes it (caching allocator: the block belongs to that stream)
                with torch.cuda.stream(self._bound_stream):
                    out = torch.empty((self.n_envs, nt, nn, slots), dtype=torch.float64, device=self.device)
        ida = (ctypes.c_int32 * len(ids))(*ids)
        _lib.check(self._L.mrx_cim_query(self._h, NODE_TYPE[node], t.data_ptr(), nt, per_env, n.data_ptr(), nn, nodes_per_env, ida,
                                         len(ids), out.data_ptr(), self._stream()), "mrx_cim_query")
        self._keep_q = (t, n)
        return out
