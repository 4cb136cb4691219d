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


# ---- the wave-cooperative decision step (maro_amd/csrc/cb_wave.h) on the 64-fiber wave emulator, in front of the general step
# exactly as mrx_cb_step launches them; forward and reverse lane order (a missing wave::sync shows up as a difference)
def _wave_cases():
    """(case, reverse lane order): every toy golden in both lane orders; of the city-sized ones — slow on 64 fibers, and replayed in
    full by the HIP build in test_gpu_citi_bike.py — one 180-station and one 800-station case, one lane order each.  Unaligned
    frames (start13 / start27) never take this path."""
    out = []
    for c in CASES:
        if "start13" in c or "start27" in c:
            continue
        if "city" not in c:
            out += [(c, False), (c, True)]
    return out + [("cb_city180_d1440_r10_ring16_half", False), ("cb_city800_d300_r20_ring12_half", True)]


@pytest.mark.parametrize("case,reverse", _wave_cases())
def test_wave_cooperative_decision_step_reproduces_reference(case, reverse):
    made = []

    def make_wave(data, kw, tt, n_envs=1):
        b = CbEmuBackend(data, n_envs=n_envs, max_actions=1, wave_decisions=True, reverse=reverse, **kw)
        b.reset(transfer_times=[tt[: b.layout.transfer_times_cap]] * n_envs)
        made.append(b)
        return CbBackendEnv(b, env=n_envs - 1)
    replay_citi_bike(make_wave, case)
    handled, general = made[0].wave_counts()
    assert handled > 0 and general > 0, (handled, general)   # both paths ran (every tick ends on the general one)
    if "city800" in case:
        assert handled > 50 * general, (handled, general)    # hundreds of decisions per decision tick stay inside the tick


# ---- ... and the GENERAL step in its wave form as well (cb::step_env_wave: plan-specialised LDS-frame builds, what
# mrx_k_cb_replay_wave runs for the envs the decision step leaves alone): lane 0 replays the events out of the LDS column, the
# station sweeps / snapshot / next-decision scan / action scope run across the lanes
@pytest.mark.parametrize("case,reverse", [(c, r) for c, r in _wave_cases() if "city800" not in c])
def test_wave_form_general_step_reproduces_reference(case, reverse):
    made = []

    def make_wave(data, kw, tt, n_envs=1):
        b = CbEmuBackend(data, n_envs=n_envs, max_actions=1, specialized=True, wave_decisions=2, reverse=reverse, **kw)
        b.reset(transfer_times=[tt[: b.layout.transfer_times_cap]] * n_envs)
        made.append(b)
        return CbBackendEnv(b, env=n_envs - 1)
    replay_citi_bike(make_wave, case)
    handled, general = made[0].wave_counts()
    assert handled > 0 and general > 0, (handled, general)
