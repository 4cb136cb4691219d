"""Python binding of tests/emu/libcim_emu.so — TEST INFRASTRUCTURE ONLY (CPU emulation of the
device code; the product never loads it)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(HERE, "libcim_emu.so")


class MrxCimConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("n_envs", "device", "start_tick", "durations", "snapshot_resolution",
                                              "max_snapshots", "max_actions", "max_stops", "decision_mode", "order_table")]


class MrxCimLayout(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_int32) for n in ("n_envs", "n_ports", "n_vessels", "frame_words", "ring_slots",
                                               "max_stops", "horizon", "frame_off_ports", "frame_off_vessels",
                                               "frame_off_full_on_ports", "frame_off_full_on_vessels",
                                               "frame_off_vessel_plans")]
                + [(n, ctypes.c_int64) for n in ("off_live", "off_ring", "off_ring_fi", "off_status", "off_tick",
                                                 "off_seed", "off_stops", "off_nstops", "off_order_prop",
                                                 "off_vessel_period", "off_orders")]
                + [(n, ctypes.c_int32) for n in ("order_row_words", "order_table_on", "order_elem_bytes", "reserved0")]
                + [("workspace_bytes", ctypes.c_int64)])


def build():
    srcs = [os.path.join(HERE, "cim_emu.cpp"), os.path.join(HERE, "wave_emu.h")] + [
        os.path.join(REPO, "maro_amd", "csrc", f) for f in ("cim_device.h", "cim_layout.h", "cim_params.h")]
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(s) for s in srcs):
        tmp = LIB + f".{os.getpid()}.tmp"   # built aside and renamed: concurrent test workers never load a half-written library
        subprocess.check_call(["g++", "-O2", "-g", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wall",
                               "-Wno-unused-function", "-shared", "-o", tmp, srcs[0]])
        os.replace(tmp, LIB)
    return LIB


def build_specialized(defines: str) -> str:
    """The emulator with the device source compiled as a plan-specialised build (MRX_SPECIALIZED + the MRXC_* text of
    mrx_cim_plan_defines): what cim_spec.hip compiles for the GPU, checked against the oracle on the CPU.  One .so per plan."""
    import hashlib
    import tempfile
    h = hashlib.sha256(defines.encode())
    for f in (os.path.join(HERE, "cim_emu.cpp"), os.path.join(HERE, "wave_emu.h"), os.path.join(REPO, "maro_amd", "csrc", "cim_device.h")):
        h.update(open(f, "rb").read())
    d = os.path.join(tempfile.gettempdir(), "maro_amd_cim_emu_spec")
    os.makedirs(d, exist_ok=True)
    so = os.path.join(d, h.hexdigest()[:20] + ".so")
    if not os.path.exists(so):
        hdr = so[:-3] + "_dims.h"
        with open(hdr, "w") as f:
            f.write(defines)
        tmp = so + f".{os.getpid()}.tmp"
        subprocess.check_call(["g++", "-O1", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wno-unused-function", "-DMRX_SPECIALIZED",
                               "-include", hdr, "-shared", "-o", tmp, os.path.join(HERE, "cim_emu.cpp")])
        os.replace(tmp, so)
    return so


_lib = None
_spec_libs = {}


def spec_lib(defines: str):
    if defines not in _spec_libs:
        _spec_libs[defines] = _declare(ctypes.CDLL(build_specialized(defines)))
    return _spec_libs[defines]


def lib():
    global _lib
    if _lib is None:
        _lib = _declare(ctypes.CDLL(build()))
    return _lib


def _declare(L):
    if True:
        L.emu_create.restype = ctypes.c_void_p
        L.emu_create.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
        L.emu_destroy.argtypes = [ctypes.c_void_p]
        L.emu_get_layout.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.emu_dump_dims.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
        L.emu_workspace.restype = ctypes.c_void_p
        L.emu_workspace.argtypes = [ctypes.c_void_p]
        L.emu_reset.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.emu_step.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 6 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.emu_query.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.emu_set_device_agent.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong]
        L.emu_set_port_history.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        L.emu_set_observation.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                          ctypes.c_void_p]
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data


class EmuBackend:
    """Batch backend with the same numpy-facing surface as the tests' GPU backend wrapper."""

    def __init__(self, topo, n_envs=1, start_tick=0, durations=100, snapshot_resolution=1, max_snapshots=None,
                 max_actions=2, reverse=False, order_table=0, decision_mode=0, specialized=False, step_mode=1, pipe_waves=3,
                 spec_obs=((), ())):
        """step_mode: 1 unsorted launch, 2 sorted launch (mrx_k_cim_schedule's order), 3 persistent pipelined kernel with
        `pipe_waves` waves (specialized=True and the order table only) — the three launch forms of mrx_cim_step."""
        self.step_mode, self.pipe_waves = step_mode, pipe_waves
        self.topo = topo
        self._cs = topo.c_struct()
        self.cfg = MrxCimConfig(n_envs, 0, start_tick, durations, snapshot_resolution, max_snapshots or 0,
                                max_actions, 0, decision_mode, order_table)
        self._L = lib()
        self._spec_obs = None
        if specialized:   # the fused observation's configuration is compiled in (spec_obs = (port attr ids, vessel attr ids)):
            from maro_amd.cim import specialize as spec   # set_observation must then be called with exactly that configuration
            self._spec_obs = (tuple(spec_obs[0]), tuple(spec_obs[1]))
            self._L = spec_lib(spec.plan_defines(self._cs, self.cfg, obs=self._spec_obs))
        err = ctypes.create_string_buffer(256)
        self._h = self._L.emu_create(ctypes.byref(self._cs), ctypes.byref(self.cfg), err, 256)
        if not self._h:
            raise RuntimeError(err.value.decode())
        self.layout = MrxCimLayout()
        self._L.emu_get_layout(self._h, ctypes.byref(self.layout))
        self.n_envs, self.max_actions, self.reverse = n_envs, max_actions, reverse
        self.max_tick = start_tick + durations
        base = self._L.emu_workspace(self._h)
        self._ws = (ctypes.c_uint8 * self.layout.workspace_bytes).from_address(base)
        self.ws = np.frombuffer(self._ws, dtype=np.uint8)

    def __del__(self):
        if getattr(self, "_h", None) and getattr(self, "_L", None) is not None:
            self._L.emu_destroy(self._h)
            self._h = None

    def view(self, off, dtype, shape):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        return self.ws[off:off + n].view(dtype).reshape(shape)

    def reset(self, seed_cmd=None, mask=None):
        sc = None if seed_cmd is None else np.ascontiguousarray(seed_cmd, np.int64)
        mk = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self._L.emu_reset(self._h, _ptr(sc), _ptr(mk), int(self.reverse))

    def step(self, actions=None, n_actions=None, mask=None, n_answered=None):
        a = None if actions is None else np.ascontiguousarray(actions, np.int32).reshape(self.n_envs, self.max_actions, 4)
        na = None if n_actions is None else np.ascontiguousarray(n_actions, np.int32)
        mk = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        nans = None if n_answered is None else np.ascontiguousarray(n_answered, np.int32)
        if not hasattr(self, "_dec"):
            self._dec = np.zeros((self.n_envs, 8) if self.cfg.decision_mode == 0 else (self.n_envs, self.layout.n_vessels, 8), np.int32)
            self._met = np.zeros((self.n_envs, 3), np.int64)
            self._done = np.zeros(self.n_envs, np.uint8)
        self._L.emu_step(self._h, _ptr(a), _ptr(na), _ptr(mk), _ptr(self._dec), _ptr(self._met), _ptr(self._done),
                       int(self.reverse), _ptr(nans), int(self.step_mode), int(self.pipe_waves))
        return self._dec.copy(), self._met.copy(), self._done.copy()

    def set_device_agent(self, actions=None, n_actions=None, counts=None, next_key=-1):
        """mrx_cim_set_device_agent: `actions` int32 [n, max_actions, 4] / `n_actions` int32 [n] / `counts` int32 [n] are written IN
        PLACE by every step (numpy arrays the caller keeps and passes to the next step)."""
        on = actions is not None
        self._agent_keep = (actions, n_actions, counts)
        self._L.emu_set_device_agent(self._h, 1 if on else 0, _ptr(actions), _ptr(n_actions), _ptr(counts), int(next_key))

    def set_observation(self, port_attr_ids, vessel_attr_ids):
        """Fused observation (mrx_cim_set_observation): returns (obs_ports [n, P, np], obs_vessel [n, nv]) written by step()."""
        pa, va = np.ascontiguousarray(port_attr_ids, np.int32), np.ascontiguousarray(vessel_attr_ids, np.int32)
        assert self._spec_obs is None or self._spec_obs == (tuple(pa.tolist()), tuple(va.tolist())), "specialised build: other observation compiled in"
        self.obs_ports = np.zeros((self.n_envs, self.layout.n_ports, len(pa)), np.float64)
        self.obs_vessel = np.zeros((self.n_envs, len(va)), np.float64)
        self._L.emu_set_observation(self._h, _ptr(pa), len(pa), _ptr(va), len(va), _ptr(self.obs_ports), _ptr(self.obs_vessel))
        return self.obs_ports, self.obs_vessel

    def set_port_history(self, port_attr_ids):
        """mrx_cim_set_port_history: int32 [n, frames, len(ids), P] written at every snapshot."""
        ids = np.ascontiguousarray(port_attr_ids, np.int32)
        frames = -(-self.cfg.durations // self.cfg.snapshot_resolution)
        self.port_history = np.zeros((self.n_envs, frames, len(ids), self.layout.n_ports), np.int32)
        self._L.emu_set_port_history(self._h, _ptr(ids), len(ids), _ptr(self.port_history), frames)
        return self.port_history

    def query(self, node_type, ticks, nodes, attrs, row_slots):
        t = np.ascontiguousarray(ticks, np.int32)
        nt = t.shape[-1]
        per_env = nt if t.ndim == 2 else 0
        n = np.ascontiguousarray(nodes, np.int32)
        nn = n.shape[-1]
        npe = nn if n.ndim == 2 else 0
        a = np.ascontiguousarray(attrs, np.int32)
        out = np.zeros((self.n_envs, nt, nn, row_slots), np.float64)
        self._L.emu_query(self._h, node_type, _ptr(t), nt, per_env, _ptr(n), nn, npe, _ptr(a), len(a), _ptr(out))
        return out
