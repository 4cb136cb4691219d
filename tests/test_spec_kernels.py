"""Plan-specialised step kernels (maro_amd/cim/specialize.py, include/maro_amd.h: mrx_cim_plan_defines): host-side pieces —
no GPU needed, hipcc cross-compiles the gfx950 code object here."""
import os
import re
import shutil

import pytest

from maro_amd import _lib
from maro_amd.cim import specialize as spec
from maro_amd.cim.topology import load_topology

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(n_envs=4, durations=100, ring=0, order_table=0, mode=0, res=1):
    return _lib.MrxCimConfig(n_envs, 0, 0, durations, res, ring, 1, 0, mode, order_table)


def test_plan_defines_cover_every_dimension_field():
    text = spec.plan_defines(load_topology("global_trade.22p_l0.8").c_struct(), _cfg(durations=1120, ring=4))
    got = {k: v for k, v in re.findall(r"#define MRXC_(\w+) (-?\d+)", text) if not k.startswith("obs_")}
    params = open(os.path.join(REPO, "maro_amd", "csrc", "cim_params.h")).read()
    fields = re.findall(r"X\((\w+)\)", params[params.index("#define MRX_CIM_DIM_FIELDS"):params.index("// Observation fused")])
    assert sorted(got) == sorted(fields) and len(fields) > 50
    assert (got["P"], got["V"], got["NT"], got["S"], got["T"], got["pregen"]) == ("22", "46", "157", "4", "1120", "1")
    # the device source reads exactly these fields through KD(), and none of them directly any more
    dev = open(os.path.join(REPO, "maro_amd", "csrc", "cim_device.h")).read()
    assert set(re.findall(r"KD\((\w+)\)", dev)) - {"f"} <= set(fields)
    assert not [f for f in fields if re.search(r"\bK\." + f + r"\b", dev)]


def test_plan_defines_include_the_fused_observation():
    cs, cfg = load_topology("toy.5p_ssddd_l0.5").c_struct(), _cfg(durations=80)
    off = spec.plan_defines(cs, cfg)
    assert "#define MRXC_obs_np 0\n" in off and "#define MRXC_obs_nv 0\n" in off
    on = spec.plan_defines(cs, cfg, obs=([1, 2, 5], [1, 3]))     # ports: empty, full, shortage; vessels: empty, remaining_space
    assert "#define MRXC_obs_np 3\n" in on and "#define MRXC_obs_nv 2\n" in on and "#define MRXC_obs_i_empty 0\n" in on
    assert f"#define MRXC_obs_pa_packed {1 | (2 << 4) | (5 << 8)}u\n" in on and "(i) == 1 ? 3 :" in on
    assert on != spec.plan_defines(cs, cfg, obs=([1, 2, 5], [3, 1])) and on.replace("obs_", "") != off.replace("obs_", "")
    dev = open(os.path.join(REPO, "maro_amd", "csrc", "cim_device.h")).read()
    assert not re.search(r"\bO\.(np|nv|pa_packed|i_empty|i_tc|va)\b", dev)   # every use goes through OD() / ODA()


def test_plan_defines_depend_on_the_plan_not_on_the_batch_size():
    cs = load_topology("toy.5p_ssddd_l0.5").c_struct()
    a = spec.plan_defines(cs, _cfg(n_envs=1, durations=80))
    assert a == spec.plan_defines(cs, _cfg(n_envs=4096, durations=80))
    assert a != spec.plan_defines(cs, _cfg(n_envs=1, durations=81))
    assert a != spec.plan_defines(cs, _cfg(n_envs=1, durations=80, mode=1))
    assert a != spec.plan_defines(load_topology("toy.4p_ssdd_l0.0").c_struct(), _cfg(n_envs=1, durations=80))


@pytest.mark.skipif(not (os.path.exists(spec.HIPCC) or shutil.which("hipcc")), reason="hipcc not available")
def test_code_object_compiles_and_is_cached(tmp_path, monkeypatch):
    monkeypatch.setattr(spec, "CACHE", str(tmp_path))
    defines = spec.plan_defines(load_topology("toy.4p_ssdd_l0.0").c_struct(), _cfg(durations=60, order_table=-1))
    with pytest.raises(KeyError):
        spec.code_object(defines, build=False)
    img = spec.code_object(defines)
    # only the kernel this configuration launches: online generator, no fused observation
    assert (img[:4] == b"\x7fELF" or img.startswith(b"__CLANG_OFFLOAD_BUNDLE__")) and b"mrx_k_cim_step" in img
    assert b"mrx_k_cim_step_obs" not in img and b"mrx_k_cim_step_tab" not in img and b"mrx_k_cim_reset" in img
    assert len([f for f in os.listdir(tmp_path) if f.endswith(".hsaco")]) == 1 and spec.code_object(defines, build=False) == img
    tab = spec.code_object(spec.plan_defines(load_topology("toy.4p_ssdd_l0.0").c_struct(), _cfg(durations=60)))
    assert b"mrx_k_cim_step_tab" in tab and b"mrx_k_cim_step_tab_obs" not in tab
    obs = spec.code_object(spec.plan_defines(load_topology("toy.4p_ssdd_l0.0").c_struct(), _cfg(durations=60), obs=([1], [1, 3])))
    assert b"mrx_k_cim_step_tab_obs" in obs
