"""citi_bike: the engine's device source (maro_amd/csrc/cb_device.h), compiled for the host, replays the vectors the
real reference produced — the CPU-side parity gate for the kernel logic (the HIP build of the same source is checked
on the GPU in test_gpu_citi_bike.py)."""
import pytest

from tests.cb_backend_adapter import CbBackendEnv
from tests.emu.cb_emu import CbEmuBackend
from tests.test_citi_bike_oracle import CASES, replay_citi_bike


def make(data, kw, tt, n_envs=3):
    b = CbEmuBackend(data, n_envs=n_envs, max_actions=1, **kw)
    b.reset(transfer_times=[tt[: b.layout.transfer_times_cap]] * n_envs)
    return CbBackendEnv(b, env=n_envs - 1)


@pytest.mark.parametrize("case", CASES)
def test_cb_device_code_reproduces_reference(case):
    replay_citi_bike(make, case)


# ---- Joint / JointWithSequentialAction (mrx_cb_step_joint), generic and plan-specialised device code
from tests.test_citi_bike_joint_oracle import JOINT_CASES, replay_citi_bike_joint  # noqa: E402


@pytest.mark.parametrize("specialized", [False, True])
@pytest.mark.parametrize("case", JOINT_CASES)
def test_cb_device_code_reproduces_reference_joint_modes(case, specialized):
    def make_joint(data, kw, tt, mode, n_envs=3):
        b = CbEmuBackend(data, n_envs=n_envs, max_actions=1, decision_mode=mode, specialized=specialized, **kw)
        b.reset(transfer_times=[tt[: b.layout.transfer_times_cap]] * n_envs)
        return CbBackendEnv(b, env=n_envs - 1)
    replay_citi_bike_joint(make_joint, case)
