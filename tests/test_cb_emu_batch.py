"""citi_bike device code on the CPU harness: batches with per-env actions and transfer times vs the oracle."""
import numpy as np
import pytest

from maro_amd.citi_bike.data import load_topology
from tests.cb_batch_check import run_batch_vs_oracle
from tests.emu.cb_emu import CbEmuBackend


@pytest.mark.parametrize("topology,kwargs,n", [
    ("toy.3s_4t", dict(durations=1440, snapshot_resolution=10), 12),
    ("toy.3s_tight", dict(durations=1100, snapshot_resolution=7, max_snapshots=9), 12),
    ("toy.3s_tight", dict(start_tick=300, durations=500, snapshot_resolution=1), 6),
    ("toy.5s_filters", dict(start_tick=13, durations=600, snapshot_resolution=10), 6),   # frames not aligned with the resolution
    ("toy.5s_filters", dict(start_tick=27, durations=500, snapshot_resolution=7, max_snapshots=5), 6),
])
def test_batch_matches_oracle(topology, kwargs, n):
    data = load_topology(topology)
    b = CbEmuBackend(data, n_envs=n, max_actions=1, **kwargs)
    steps = run_batch_vs_oracle(b, data, kwargs, seeds=np.arange(n) + 11, episodes=2)
    assert steps > 20


def test_masked_envs_do_not_move():
    data = load_topology("toy.3s_4t")
    b = CbEmuBackend(data, n_envs=4, durations=200)
    b.reset(transfer_times=np.full((4, 8), 20))
    mask = np.array([1, 0, 1, 0], np.uint8)
    d0, _, _, _ = b.step()
    t0 = b.hdr()[0].copy()
    b.step(mask=mask)
    t1 = b.hdr()[0]
    assert (t1[[1, 3]] == t0[[1, 3]]).all()


@pytest.mark.parametrize("budget,specialized", [(1, False), (7, True), (40, False)])
def test_bounded_steps_do_not_change_trajectories(budget, specialized):
    """mrx_cb_set_step_budget: envs that run out of budget report "no decision yet" and continue in the next call."""
    import functools

    from tests.cb_batch_check import run_bounded_vs_oracle
    from tests.emu.cb_emu import CbEmuBackend
    data = load_topology("toy.5s_6t")
    kw = dict(durations=700, snapshot_resolution=5)
    B = functools.partial(CbEmuBackend, specialized=True) if specialized else CbEmuBackend
    b = B(data, n_envs=6, max_actions=1, **kw)
    calls, unready = run_bounded_vs_oracle(b, data, kw, seeds=np.arange(6) + 3, budget=budget)
    assert unready > 0 and calls > 50


@pytest.mark.parametrize("kwargs", [dict(start_tick=13, durations=600, snapshot_resolution=10), dict(start_tick=27, durations=500, snapshot_resolution=7, max_snapshots=5)])
def test_batch_matches_oracle_specialized_unaligned_frames(kwargs):
    """The LDS-frame build with the trip-window tags in the LDS column and the computed snapshot tick, start_tick % resolution != 0."""
    import functools
    data = load_topology("toy.5s_filters")
    b = functools.partial(CbEmuBackend, specialized=True)(data, n_envs=5, max_actions=1, **kwargs)
    run_batch_vs_oracle(b, data, kwargs, seeds=np.arange(5) + 11, episodes=2)


@pytest.mark.parametrize("topology,kwargs,n", [
    ("toy.5s_filters", dict(durations=700, snapshot_resolution=10), 6),
    ("toy.3s_tight", dict(durations=900, snapshot_resolution=3, max_snapshots=9), 8),
    ("city.180s", dict(durations=260, snapshot_resolution=20, max_snapshots=12), 3),
])
def test_batch_matches_oracle_with_wave_cooperative_decisions(topology, kwargs, n):
    """Per-env actions and transfer times, two episodes: the wave-cooperative decision step (cb_wave.h) in front of the general
    step, as mrx_cb_step runs them."""
    data = load_topology(topology)
    b = CbEmuBackend(data, n_envs=n, max_actions=1, wave_decisions=True, **kwargs)
    steps = run_batch_vs_oracle(b, data, kwargs, seeds=np.arange(n) + 5, episodes=2)
    handled, general = b.wave_counts()
    assert steps > 20 and handled > 0 and general > 0


@pytest.mark.parametrize("topology,kwargs,n,budget", [
    ("toy.5s_filters", dict(durations=700, snapshot_resolution=10), 5, 0),
    ("toy.3s_tight", dict(durations=900, snapshot_resolution=3, max_snapshots=9), 6, 7),
])
def test_batch_matches_oracle_with_wave_form_general_step(topology, kwargs, n, budget):
    """Plan-specialised LDS-frame build: decision step AND general step in their wave forms (mrx_k_cb_step_wave +
    mrx_k_cb_replay_wave), with and without a step budget."""
    from tests.cb_batch_check import run_bounded_vs_oracle
    data = load_topology(topology)
    b = CbEmuBackend(data, n_envs=n, max_actions=1, specialized=True, wave_decisions=2, **kwargs)
    if budget:
        calls, unready = run_bounded_vs_oracle(b, data, kwargs, seeds=np.arange(n) + 9, budget=budget)
        assert unready > 0
    else:
        assert run_batch_vs_oracle(b, data, kwargs, seeds=np.arange(n) + 9, episodes=2) > 20
    handled, general = b.wave_counts()
    assert handled > 0 and general > 0


@pytest.mark.parametrize("specialized,wave", [(False, 0), (False, 1), (True, 2)])
def test_env_major_layout_on_a_small_topology(monkeypatch, specialized, wave):
    """CbParams::aos (per-env state [env][words] instead of [word][env]) is chosen from 96 stations on; forced here on a toy so
    that every kernel form meets it on the CPU: general step, wave-cooperative decision step, wave-form general step."""
    monkeypatch.setenv("MRX_CB_AOS", "1")
    data = load_topology("toy.5s_filters")
    kwargs = dict(durations=600, snapshot_resolution=10, max_snapshots=7)
    b = CbEmuBackend(data, n_envs=5, max_actions=1, specialized=specialized, wave_decisions=wave, **kwargs)
    assert b.layout.env_major == 1
    assert run_batch_vs_oracle(b, data, kwargs, seeds=np.arange(5) + 21, episodes=2) > 20
