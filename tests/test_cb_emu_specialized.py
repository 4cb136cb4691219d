"""citi_bike, plan-specialised build of the device source on the host (tests/emu/cb_emu.py::build_specialized): the CD()
constants and — for frames of at most 128 words — the register-resident frame (MRX_CB_REGFRAME, LvRef select chains) replay
the reference's vectors and random data sets exactly like the generic build.  CPU-side gate for what cb_spec.hip compiles."""
import functools

import pytest

from tests.cb_backend_adapter import CbBackendEnv
from tests.emu.cb_emu import CbEmuBackend
from tests.test_citi_bike_oracle import CASES, replay_citi_bike

SpecBackend = functools.partial(CbEmuBackend, specialized=True)


def make(data, kw, tt, n_envs=3):
    b = SpecBackend(data, n_envs=n_envs, max_actions=1, **kw)
    b.reset(transfer_times=[tt[: b.layout.transfer_times_cap]] * n_envs)
    return CbBackendEnv(b, env=n_envs - 1)


@pytest.mark.parametrize("case", CASES[::2])
def test_specialized_cb_device_code_reproduces_reference(case):
    replay_citi_bike(make, case)


@pytest.mark.parametrize("case_seed", [1, 5, 22, 36, 58, 77])   # 3..40 stations: register-frame and HBM-frame builds
def test_specialized_cb_device_code_on_random_data(case_seed):
    from tests.fuzz_citi_bike import run_case
    assert run_case(case_seed, backend=SpecBackend) >= 0


def test_register_frame_is_what_small_plans_compile():
    import re

    from maro_amd.cim import specialize as spec
    from maro_amd.citi_bike.abi import MrxCbConfig, topology_struct
    from maro_amd.citi_bike.data import load_topology
    fw = {}
    for name in ("toy.3s_4t", "toy.5s_6t"):
        ts, keep = topology_struct(load_topology(name))
        d = spec.plan_defines(ts, MrxCbConfig(64, 0, 0, 500, 10, 0, 1, 20, 0), "citi_bike")
        fw[name] = int(re.search(r"#define MRXC_FW (\d+)", d).group(1))
    assert fw == {"toy.3s_4t": 24, "toy.5s_6t": 40} and max(fw.values()) <= 128   # both below the MRX_CB_REGFRAME threshold
