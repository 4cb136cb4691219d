"""Loader of the HIP extension (maro_amd/csrc/libmaro_amd.so) — the ONLY compute path.

There is deliberately no CPU / PyTorch fallback: if the shared library is missing or cannot be
loaded the import of any engine class fails loudly.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MARO_AMD_LIB") or os.path.join(_HERE, "csrc", "libmaro_amd.so")   # ($MARO_AMD_LIB: profiling builds, tools/)


class MrxCimConfig(ctypes.Structure):
    """ctypes mirror of ``struct mrx_cim_config`` (include/maro_amd.h)."""

    _fields_ = [(n, ctypes.c_int32) for n in ("n_envs", "device", "start_tick", "durations", "snapshot_resolution",
                                              "max_snapshots", "max_actions", "max_stops", "decision_mode", "order_table")]


class MrxCimLayout(ctypes.Structure):
    """ctypes mirror of ``struct mrx_cim_layout`` (include/maro_amd.h)."""

    _fields_ = ([(n, ctypes.c_int32) for n in ("n_envs", "n_ports", "n_vessels", "frame_words", "ring_slots",
                                               "max_stops", "horizon", "frame_off_ports", "frame_off_vessels",
                                               "frame_off_full_on_ports", "frame_off_full_on_vessels",
                                               "frame_off_vessel_plans")]
                + [(n, ctypes.c_int64) for n in ("off_live", "off_ring", "off_ring_fi", "off_status", "off_tick",
                                                 "off_seed", "off_stops", "off_nstops", "off_order_prop",
                                                 "off_vessel_period", "off_orders")]
                + [(n, ctypes.c_int32) for n in ("order_row_words", "order_table_on", "order_elem_bytes", "reserved0")]
                + [("workspace_bytes", ctypes.c_int64)])


class MrxCimDqnModel(ctypes.Structure):
    """ctypes mirror of ``struct mrx_cim_dqn_model`` (include/maro_amd.h)."""

    _fields_ = [("n_nets", ctypes.c_int32), ("n_layers", ctypes.c_int32), ("dims", ctypes.c_int32 * 9),
                ("dueling", ctypes.c_int32), ("n_actions", ctypes.c_int32), ("negative_slope", ctypes.c_float),
                ("epsilon", ctypes.c_float), ("look_back", ctypes.c_int32), ("n_port_attrs", ctypes.c_int32),
                ("port_attrs", ctypes.c_int32 * 8), ("n_vessel_attrs", ctypes.c_int32), ("vessel_attrs", ctypes.c_int32 * 8),
                ("action_space", ctypes.c_double * 32), ("d_weights", ctypes.c_void_p)]


class MrxCimSamplerCache(ctypes.Structure):
    """ctypes mirror of ``struct mrx_cim_sampler_cache`` (include/maro_amd.h)."""

    _fields_ = ([(n, ctypes.c_int32) for n in ("n_envs", "n_ports", "state_dim", "cap", "state_f64", "window", "frames", "reserved0")]
                + [("fulfillment_factor", ctypes.c_double), ("shortage_factor", ctypes.c_double)]
                + [(n, ctypes.c_void_p) for n in ("d_decay", "d_eoe", "d_head", "d_tail", "d_last", "d_prev_j", "d_prev_active", "d_interactions", "c_tick",
                                                  "c_agent", "c_state", "c_action", "c_env_action", "c_terminal", "c_next_state", "c_next_agent_state",
                                                  "d_port_history")])


EXPORTS = ("mrx_last_error", "mrx_version", "mrx_cim_workspace_bytes", "mrx_cim_create", "mrx_cim_destroy",
           "mrx_cim_get_layout", "mrx_cim_reset", "mrx_cim_step", "mrx_cim_step_joint", "mrx_cim_query", "mrx_cim_attr_id",
           "mrx_cim_attr_slots", "mrx_cim_random_policy", "mrx_cim_set_device_agent", "mrx_cim_set_observation", "mrx_cim_set_step_mode", "mrx_cim_set_port_history", "mrx_cim_dqn_net_floats",
           "mrx_cim_dqn_pack_net", "mrx_cim_dqn_scratch_bytes", "mrx_cim_dqn_act", "mrx_cim_sampler_record", "mrx_cim_sampler_emit", "mrx_cim_collect_steps", "mrx_cim_sampler_finalize", "mrx_cim_sampler_emit_all", "mrx_cim_plan_defines", "mrx_cim_load_step_kernels", "mrx_cim_read_kernel_global",
           # include/maro_amd_citi_bike.h
           "mrx_cb_workspace_bytes", "mrx_cb_create", "mrx_cb_destroy", "mrx_cb_get_layout", "mrx_cb_reset", "mrx_cb_step",
           "mrx_cb_query", "mrx_cb_random_policy", "mrx_cb_attr_id", "mrx_cb_attr_slots", "mrx_cb_plan_defines",
           "mrx_cb_load_step_kernels", "mrx_cb_set_lanes_per_wave", "mrx_cb_set_step_budget", "mrx_cb_step_joint", "mrx_cb_set_wave_decisions", "mrx_cb_set_replay_overlap", "mrx_cb_set_replay_period", "mrx_cb_set_observation", "mrx_cb_observation_rows")

_lib = None


class ExtensionMissingError(ImportError):
    pass


def load() -> ctypes.CDLL:
    """Load libmaro_amd.so and declare every entry point of include/maro_amd.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ExtensionMissingError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c \"import __graft_entry__ as g; "
            f"g.build()\"` (hipcc --offload-arch=gfx950). maro_amd has no CPU fallback.")
    # PyTorch wheels bundle their own libamdhip64; it must be the one this process binds (device memory and streams
    # come from torch), so torch is imported before the extension pulls in a second copy from /opt/rocm.
    import torch  # noqa: F401
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
    L.mrx_last_error.restype = ctypes.c_char_p
    L.mrx_version.restype = ctypes.c_char_p
    L.mrx_cim_workspace_bytes.restype = i64
    L.mrx_cim_workspace_bytes.argtypes = [vp, vp]
    L.mrx_cim_create.restype = i32
    L.mrx_cim_create.argtypes = [vp, vp, vp, i64, ctypes.POINTER(vp)]
    L.mrx_cim_destroy.restype = i32
    L.mrx_cim_destroy.argtypes = [vp]
    L.mrx_cim_get_layout.restype = i32
    L.mrx_cim_get_layout.argtypes = [vp, vp]
    L.mrx_cim_reset.restype = i32
    L.mrx_cim_reset.argtypes = [vp, vp, vp, vp]
    L.mrx_cim_step.restype = i32
    L.mrx_cim_step.argtypes = [vp] * 8
    L.mrx_cim_step_joint.restype = i32
    L.mrx_cim_step_joint.argtypes = [vp] * 9
    L.mrx_cim_query.restype = i32
    L.mrx_cim_query.argtypes = [vp, i32, vp, i32, i32, vp, i32, i32, vp, i32, vp, vp]
    L.mrx_cim_random_policy.restype = i32
    L.mrx_cim_random_policy.argtypes = [vp, vp, i64, vp, vp, vp, vp]
    L.mrx_cim_set_device_agent.restype = i32
    L.mrx_cim_set_device_agent.argtypes = [vp, i32, vp, vp, vp, i64]
    L.mrx_cim_set_observation.restype = i32
    L.mrx_cim_set_observation.argtypes = [vp, vp, i32, vp, i32, vp, vp]
    L.mrx_cim_set_port_history.restype = i32
    L.mrx_cim_set_port_history.argtypes = [vp, vp, i32, vp, i64]
    L.mrx_cim_set_step_mode.restype = i32
    L.mrx_cim_set_step_mode.argtypes = [vp, i32]
    L.mrx_cim_plan_defines.restype = i64
    L.mrx_cim_plan_defines.argtypes = [vp, vp, vp, i32, vp, i32, ctypes.c_char_p, i64]
    L.mrx_cim_load_step_kernels.restype = i32
    L.mrx_cim_load_step_kernels.argtypes = [vp, vp, i64, ctypes.c_char_p]
    L.mrx_cim_read_kernel_global.restype = i32
    L.mrx_cim_read_kernel_global.argtypes = [vp, ctypes.c_char_p, vp, i64, i32]
    L.mrx_cim_dqn_net_floats.restype = i64
    L.mrx_cim_dqn_net_floats.argtypes = [vp]
    L.mrx_cim_dqn_pack_net.restype = i32
    L.mrx_cim_dqn_pack_net.argtypes = [vp, vp, vp, vp]
    L.mrx_cim_dqn_scratch_bytes.restype = i64
    L.mrx_cim_dqn_scratch_bytes.argtypes = [vp]
    L.mrx_cim_dqn_act.restype = i32
    L.mrx_cim_dqn_act.argtypes = [vp] * 11
    L.mrx_cim_sampler_record.restype = i32
    L.mrx_cim_sampler_record.argtypes = [i32] * 7 + [vp] * 20 + [i32, vp]
    L.mrx_cim_sampler_emit.restype = i32
    L.mrx_cim_sampler_emit.argtypes = [i32] * 7 + [ctypes.c_double] * 2 + [vp] * 24 + [i32, vp]
    L.mrx_cim_collect_steps.restype = i32
    L.mrx_cim_collect_steps.argtypes = [vp] * 9 + [i32, vp]
    L.mrx_cim_sampler_finalize.restype = i32
    L.mrx_cim_sampler_finalize.argtypes = [vp] * 7
    L.mrx_cim_sampler_emit_all.restype = i32
    L.mrx_cim_sampler_emit_all.argtypes = [vp] * 15
    L.mrx_cim_attr_id.restype = i32
    L.mrx_cim_attr_id.argtypes = [i32, ctypes.c_char_p]
    L.mrx_cim_attr_slots.restype = i32
    L.mrx_cim_attr_slots.argtypes = [vp, i32, i32]
    # ---- include/maro_amd_citi_bike.h
    L.mrx_cb_workspace_bytes.restype = i64
    L.mrx_cb_workspace_bytes.argtypes = [vp, vp]
    L.mrx_cb_create.restype = i32
    L.mrx_cb_create.argtypes = [vp, vp, vp, i64, ctypes.POINTER(vp)]
    L.mrx_cb_destroy.restype = i32
    L.mrx_cb_destroy.argtypes = [vp]
    L.mrx_cb_get_layout.restype = i32
    L.mrx_cb_get_layout.argtypes = [vp, vp]
    L.mrx_cb_reset.restype = i32
    L.mrx_cb_reset.argtypes = [vp, vp, i32, vp, vp]
    L.mrx_cb_step.restype = i32
    L.mrx_cb_step.argtypes = [vp] * 9
    L.mrx_cb_query.restype = i32
    L.mrx_cb_query.argtypes = [vp, i32, vp, i32, i32, vp, i32, i32, vp, i32, vp, vp]
    L.mrx_cb_set_lanes_per_wave.restype = i32
    L.mrx_cb_set_lanes_per_wave.argtypes = [vp, i32]
    L.mrx_cb_step_joint.restype = i32
    L.mrx_cb_step_joint.argtypes = [vp] * 10
    L.mrx_cb_set_step_budget.restype = i32
    L.mrx_cb_set_step_budget.argtypes = [vp, i32]
    L.mrx_cb_set_observation.restype = i32
    L.mrx_cb_set_observation.argtypes = [vp, vp, i32, vp]
    L.mrx_cb_observation_rows.restype = i32
    L.mrx_cb_observation_rows.argtypes = [vp]
    L.mrx_cb_set_wave_decisions.restype = i32
    L.mrx_cb_set_wave_decisions.argtypes = [vp, i32]
    L.mrx_cb_set_replay_overlap.restype = i32
    L.mrx_cb_set_replay_overlap.argtypes = [vp, i32]
    L.mrx_cb_set_replay_period.restype = i32
    L.mrx_cb_set_replay_period.argtypes = [vp, i32, i32]
    L.mrx_cb_random_policy.restype = i32
    L.mrx_cb_random_policy.argtypes = [vp, vp, vp, i64, vp, vp, vp, vp]
    L.mrx_cb_plan_defines.restype = i64
    L.mrx_cb_plan_defines.argtypes = [vp, vp, ctypes.c_char_p, i64]
    L.mrx_cb_load_step_kernels.restype = i32
    L.mrx_cb_load_step_kernels.argtypes = [vp, vp, i64, ctypes.c_char_p]
    L.mrx_cb_attr_id.restype = i32
    L.mrx_cb_attr_id.argtypes = [i32, ctypes.c_char_p]
    L.mrx_cb_attr_slots.restype = i32
    L.mrx_cb_attr_slots.argtypes = [vp, i32, i32]
    _lib = L
    return L


_SRC_HASH = {}


def source_hash(scope: str = "") -> str:
    """sha256 (first 16 hex digits) over the library's sources (maro_amd/csrc/*.h, *.hip, include/*.h): the identity of the kernels
    that are NOT plan-specialised code objects (DQN forward, sampler kernels, queries ...).  bench.py stamps its lines with it and
    only uses a PMC record of the same sources.  `scope` "cim" / "cb": only that scenario's translation unit — its own files
    (cim_* + maro_amd.h, cb_* + maro_amd_citi_bike.h) and the shared wave.h — so that a change to one scenario's kernels does not
    retire the other scenario's measurements."""
    if scope not in _SRC_HASH:
        import hashlib
        h = hashlib.sha256()
        inc = os.path.join(os.path.dirname(_HERE), "include")
        files = sorted(os.path.join(d, f) for d in (os.path.join(_HERE, "csrc"), inc) for f in os.listdir(d) if f.endswith((".h", ".hip")))
        own = {"cim": ("cim_", "maro_amd.h", "wave.h"), "cb": ("cb_", "maro_amd_citi_bike.h", "wave.h")}.get(scope)
        for f in files:
            b = os.path.basename(f)
            if own is not None and not (b.startswith(own[0]) or b in own[1:]):
                continue
            h.update(b.encode())
            with open(f, "rb") as fp:
                h.update(fp.read())
        _SRC_HASH[scope] = h.hexdigest()[:16]
    return _SRC_HASH[scope]


def check(rc: int, what: str):
    if rc < 0:
        msg = load().mrx_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed ({rc}): {msg}")
    return rc
