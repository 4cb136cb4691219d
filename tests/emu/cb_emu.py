"""Python binding of tests/emu/libcb_emu.so — TEST INFRASTRUCTURE ONLY (the citi_bike device code compiled for
the host; the product never loads it)."""
import ctypes
import os
import subprocess

import numpy as np

from maro_amd.citi_bike.abi import MrxCbConfig, MrxCbLayout, topology_struct

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(HERE, "libcb_emu.so")


def build():
    srcs = [os.path.join(HERE, "cb_emu.cpp")] + [os.path.join(REPO, "maro_amd", "csrc", f) for f in
                                                 ("cb_device.h", "cb_layout.h", "cb_params.h", "cb_wave.h")] + [
        os.path.join(REPO, "include", "maro_amd_citi_bike.h"), os.path.join(HERE, "wave_emu.h")]
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(s) for s in srcs):
        tmp = LIB + f".{os.getpid()}.tmp"   # built aside and renamed: concurrent test workers never load a half-written library
        subprocess.check_call(["g++", "-O2", "-g", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wall",
                               "-Wno-unused-function", "-shared", "-o", tmp, srcs[0]])
        os.replace(tmp, LIB)
    return LIB


def build_specialized(defines: str) -> str:
    """The same harness with the device source compiled as a plan-specialised build (MRX_SPECIALIZED + the plan's MRXC_* text
    of mrx_cb_plan_defines): CD() constants, and for small frames the register-resident frame (MRX_CB_REGFRAME / LvRef) — so
    that path is checked against the oracle on the CPU too.  One .so per plan, cached in a temp dir."""
    import hashlib
    import tempfile
    key = hashlib.sha256(defines.encode() + b"".join(open(os.path.join(REPO, "maro_amd", "csrc", f), "rb").read() for f in
                                                     ("cb_device.h", "cb_layout.h", "cb_params.h", "cb_wave.h"))
                         + open(os.path.join(HERE, "cb_emu.cpp"), "rb").read() + open(os.path.join(HERE, "wave_emu.h"), "rb").read()).hexdigest()[:20]
    d = os.path.join(tempfile.gettempdir(), "maro_amd_cb_emu_spec")
    os.makedirs(d, exist_ok=True)
    so = os.path.join(d, key + ".so")
    if not os.path.exists(so):
        hdr = os.path.join(d, key + "_dims.h")
        with open(hdr, "w") as f:
            f.write(defines)
        tmp = so + f".{os.getpid()}.tmp"
        subprocess.check_call(["g++", "-O2", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wno-unused-function", "-DMRX_SPECIALIZED",
                               "-include", hdr, "-shared", "-o", tmp, os.path.join(HERE, "cb_emu.cpp")])
        os.replace(tmp, so)
    return so


_lib = None
_spec_libs = {}


def _declare(L):
    vp, i32 = ctypes.c_void_p, ctypes.c_int
    L.cb_emu_create.restype = vp
    L.cb_emu_create.argtypes = [vp, vp, ctypes.c_char_p, i32]
    L.cb_emu_destroy.argtypes = [vp]
    L.cb_emu_get_layout.argtypes = [vp, vp]
    L.cb_emu_workspace.restype = vp
    L.cb_emu_workspace.argtypes = [vp]
    L.cb_emu_reset.argtypes = [vp, vp, i32, vp]
    L.cb_emu_step.argtypes = [vp] * 8
    L.cb_emu_step_joint.argtypes = [vp] * 9
    L.cb_emu_query.argtypes = [vp, i32, vp, i32, i32, vp, i32, i32, vp, i32, vp]
    L.cb_emu_random_policy.argtypes = [vp, vp, vp, ctypes.c_int64, vp, vp]
    L.cb_emu_set_wave_decisions.argtypes = [vp, i32, i32]
    L.cb_emu_set_replay_overlap.argtypes = [vp, i32]
    L.cb_emu_rank.argtypes = [vp, i32, i32, vp, vp, vp]
    L.cb_emu_set_pool_stage.argtypes = [vp, i32]
    L.cb_emu_set_replay_period.argtypes = [vp, i32]
    L.cb_emu_set_observation.argtypes = [vp, vp, i32, vp]
    L.cb_emu_wave_handled.restype = ctypes.c_long
    L.cb_emu_wave_handled.argtypes = [vp]
    L.cb_emu_wave_general.restype = ctypes.c_long
    L.cb_emu_wave_general.argtypes = [vp]
    return L


def spec_lib(defines: str):
    if defines not in _spec_libs:
        _spec_libs[defines] = _declare(ctypes.CDLL(build_specialized(defines)))
    return _spec_libs[defines]


def lib():
    global _lib
    if _lib is None:
        _lib = _declare(ctypes.CDLL(build()))
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data


class CbEmuBackend:
    """numpy-facing batch backend; tests/cb_backend_adapter.py gives the GPU engine the same surface."""

    def __init__(self, data, n_envs=1, start_tick=0, durations=100, snapshot_resolution=1, max_snapshots=None, max_actions=1,
                 delivery_capacity=0, transfer_times_cap=0, specialized=False, decision_mode=0, wave_decisions=False, reverse=False):
        self.data = data
        if not delivery_capacity:   # same default as maro_amd.citi_bike.engine.CitiBikeBatchEngine
            delivery_capacity = data.n_stations * (int((data.time_mean + 6 * data.time_std) / max(data.resolution, 1)) + 2) + 4
        self._ts, self._keep = topology_struct(data)
        self.cfg = MrxCbConfig(n_envs, 0, start_tick, durations, snapshot_resolution, max_snapshots or 0, max_actions,
                               delivery_capacity, transfer_times_cap, decision_mode)
        self.decision_mode = decision_mode
        self._L = lib()
        if specialized:   # the plan's constants come from the product library's host-only mrx_cb_plan_defines
            from maro_amd.cim import specialize as spec
            self._L = spec_lib(spec.plan_defines(self._ts, self.cfg, "citi_bike"))
        err = ctypes.create_string_buffer(256)
        self._h = self._L.cb_emu_create(ctypes.byref(self._ts), ctypes.byref(self.cfg), err, 256)
        if not self._h:
            raise RuntimeError(err.value.decode())
        self.layout = MrxCbLayout()
        self._L.cb_emu_get_layout(self._h, ctypes.byref(self.layout))
        self.n_envs, self.max_actions = n_envs, max_actions
        self.start_tick, self.max_tick, self.res = start_tick, start_tick + durations, snapshot_resolution
        base = self._L.cb_emu_workspace(self._h)
        self._ws = (ctypes.c_uint8 * self.layout.workspace_bytes).from_address(base)
        self.ws = np.frombuffer(self._ws, dtype=np.uint8)
        rows = (data.n_stations,) if decision_mode else ()
        self._dec = np.zeros((n_envs,) + rows + (8,), np.int32)
        self._scope = np.zeros((n_envs,) + rows + (self.layout.scope_cap, 2), np.int32)
        self._met = np.zeros((n_envs, 3), np.int64)
        self._done = np.zeros(n_envs, np.uint8)
        self._wave = int(wave_decisions)
        if wave_decisions:   # steps go through the wave-cooperative decision step first (cb_wave.h on the 64-fiber wave emulator);
            # 2 (specialised LDS-frame builds): the general step in its wave form as well (cb::step_env_wave)
            self._L.cb_emu_set_wave_decisions(ctypes.c_void_p(self._h), int(wave_decisions), int(reverse))

    def rank(self, n, mode, v, key):
        """cb::cbw_rank (the wave kernels' counting rank) on the 64-fiber emulator: positions >= n are ignored."""
        vv, kk = np.zeros(128, np.int32), np.full(128, -1, np.int32)
        vv[:n], kk[:n] = v, key
        out = np.zeros(128, np.int32)
        self._L.cb_emu_rank(ctypes.c_void_p(self._h), int(n), int(mode), _ptr(vv), _ptr(kk), _ptr(out))
        return out[:n]

    def set_pool_stage(self, entries):
        """K.pool_stage of the wave replay step (env-major builds): how many pool entries from the ring's head on are read out of the LDS
        copy (0: none, the bucket table neither; the product passes CB_POOL_STAGE = 256)."""
        self._L.cb_emu_set_pool_stage(ctypes.c_void_p(self._h), int(entries))

    def set_replay_period(self, n):
        """mrx_cb_set_replay_period on the harness (wave_decisions=2): the general step on every n-th call, deferred envs in between."""
        self._L.cb_emu_set_replay_period(ctypes.c_void_p(self._h), int(n))

    def set_replay_overlap(self, on=True):
        """mrx_cb_set_replay_overlap on the harness: classify every env first, then the two wave kernels on disjoint envs (the default)."""
        self._L.cb_emu_set_replay_overlap(ctypes.c_void_p(self._h), int(bool(on)))

    def set_observation(self, attr_ids):
        """mrx_cb_set_observation: float64 [n, rows, len(ids)] written by every step (rows: S, or scope_cap on the wave path)."""
        ids = np.ascontiguousarray(attr_ids, np.int32)
        rows = self.layout.scope_cap if self._wave else self.data.n_stations
        self.obs = np.zeros((self.n_envs, rows, len(ids)), np.float64)
        self._L.cb_emu_set_observation(ctypes.c_void_p(self._h), _ptr(ids), len(ids), _ptr(self.obs))
        return self.obs

    def wave_counts(self):
        """(env-steps handled by the wave-cooperative decision step, env-steps that went to the general path)"""
        return int(self._L.cb_emu_wave_handled(ctypes.c_void_p(self._h))), int(self._L.cb_emu_wave_general(ctypes.c_void_p(self._h)))

    def __del__(self):
        if getattr(self, "_h", None) and getattr(self, "_L", None) is not None:
            self._L.cb_emu_destroy(self._h)
            self._h = None

    def view(self, off, words):
        """int32 [words, n_envs] view of a per-env SoA array."""
        st = self.layout.env_stride
        if self.layout.env_major:   # [env][words] in memory: the same [word, env] indexing through a transpose
            return self.ws[off:off + words * st * 4].view(np.int32).reshape(st, words)[:self.n_envs].T
        return self.ws[off:off + words * st * 4].view(np.int32).reshape(words, st)[:, :self.n_envs]

    def reset(self, transfer_times=None, mask=None):
        tt = None if transfer_times is None else np.ascontiguousarray(transfer_times, np.int32).reshape(self.n_envs, -1)
        mk = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self._L.cb_emu_reset(self._h, _ptr(tt), 0 if tt is None else tt.shape[1], _ptr(mk))

    def set_step_budget(self, max_records):
        self._L.cb_emu_set_step_budget(ctypes.c_void_p(self._h), int(max_records))

    def step(self, actions=None, n_actions=None, mask=None):
        a = None if actions is None else np.ascontiguousarray(actions, np.int32).reshape(self.n_envs, self.max_actions, 3)
        na = None if n_actions is None else np.ascontiguousarray(n_actions, np.int32)
        mk = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self._L.cb_emu_step(self._h, _ptr(a), _ptr(na), _ptr(mk), _ptr(self._dec), _ptr(self._scope), _ptr(self._met), _ptr(self._done))
        return self._dec.copy(), self._scope.copy(), self._met.copy(), self._done.copy()

    def step_joint(self, actions=None, n_actions=None, n_answered=None, mask=None):
        """actions [n, S, A, 3], n_actions [n, S], n_answered [n] (Joint modes)."""
        S = self.data.n_stations
        a = None if actions is None else np.ascontiguousarray(actions, np.int32).reshape(self.n_envs, S, self.max_actions, 3)
        na = None if n_actions is None else np.ascontiguousarray(n_actions, np.int32).reshape(self.n_envs, S)
        nans = None if n_answered is None else np.ascontiguousarray(n_answered, np.int32)
        mk = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self._L.cb_emu_step_joint(self._h, _ptr(a), _ptr(na), _ptr(nans), _ptr(mk), _ptr(self._dec), _ptr(self._scope), _ptr(self._met), _ptr(self._done))
        return self._dec.copy(), self._scope.copy(), self._met.copy(), self._done.copy()

    def random_policy(self, dec, scope, step):
        a = np.zeros((self.n_envs, self.max_actions, 3), np.int32)
        na = np.zeros(self.n_envs, np.int32)
        self._L.cb_emu_random_policy(self._h, _ptr(np.ascontiguousarray(dec, np.int32)), _ptr(np.ascontiguousarray(scope, np.int32)),
                                   int(step), _ptr(a), _ptr(na))
        return a, na

    def query(self, node_type, ticks, nodes, attrs, row_slots):
        t = np.ascontiguousarray(ticks, np.int32)
        nt, per_env = t.shape[-1], (t.shape[-1] if t.ndim == 2 else 0)
        n = np.ascontiguousarray(nodes, np.int32)
        nn, npe = n.shape[-1], (n.shape[-1] if n.ndim == 2 else 0)
        a = np.ascontiguousarray(attrs, np.int32)
        out = np.zeros((self.n_envs, nt, nn, row_slots), np.float64)
        self._L.cb_emu_query(self._h, node_type, _ptr(t), nt, per_env, _ptr(n), nn, npe, _ptr(a), len(a), _ptr(out))
        return out

    def hdr(self):
        return self.view(self.layout.off_hdr, 16)

    def ring_fi(self):
        return self.view(self.layout.off_ring_fi, self.layout.ring_slots)
