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


@pytest.mark.parametrize("overlap", [True, False])
@pytest.mark.parametrize("topology,kwargs,n,budget", [
    ("toy.5s_filters", dict(durations=700, snapshot_resolution=10), 5, 0),
    ("toy.3s_tight", dict(durations=900, snapshot_resolution=3, max_snapshots=9), 6, 7),
])
def test_batch_matches_oracle_with_wave_form_general_step(topology, kwargs, n, budget, overlap):
    """Plan-specialised LDS-frame build: decision step AND general step in their wave forms (mrx_k_cb_step_wave +
    mrx_k_cb_replay_wave), with and without a step budget; with the split mrx_cb_step makes by default (mrx_k_cb_classify over every
    env first, then the two kernels on disjoint envs: mrx_cb_set_replay_overlap) and with the replay kernel after the in-tick one."""
    from tests.cb_batch_check import run_bounded_vs_oracle
    data = load_topology(topology)
    b = CbEmuBackend(data, n_envs=n, max_actions=1, specialized=True, wave_decisions=2, **kwargs)
    b.set_replay_overlap(overlap)
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


@pytest.mark.parametrize("topology,specialized,wave", [("toy.5s_6t", False, 0), ("toy.5s_6t", True, 0), ("city.180s", True, 2)])
def test_fused_observation_equals_the_query_on_the_emulator(topology, specialized, wave):
    """mrx_cb_set_observation on the host-compiled device code: every station's row (lane path, generic and LDS-frame builds) or the
    rows of the action scope's stations (wave-cooperative kernels, forward and reverse lane order) equal the snapshot query of the
    decision's frame at every step, with bounded steps on the way (rows of envs without a decision are zeros)."""
    from maro_amd.citi_bike.abi import draw_transfer_times
    from maro_amd.citi_bike.data import load_topology
    from tests.emu.cb_emu import CbEmuBackend
    data = load_topology(topology)
    n, attrs = 3, [0, 1, 2, 3, 4, 10, 13, 6]      # bikes, shortage, trip_requirement, fulfillment, capacity, extra_cost, min_bikes, weekday
    lv_row = {0: 0, 1: 1, 2: 2, 3: 3, 10: 4, 13: 7}    # station attribute id -> live-frame row (cb_params.h: SA_* -> LV_*)

    def expected(be, e, nodes, t):
        """The slice straight from the env's live frame in the workspace (the decision's frame IS the live frame) and the shared tables
        — not through the harness's query, whose LDS-frame host build reads a static stand-in of the LDS column."""
        live = be.view(be.layout.off_live, be.layout.frame_words)
        S = data.n_stations
        out = np.zeros((len(nodes), len(attrs)))
        for r, st in enumerate(nodes):
            if st < 0:
                continue
            for c, a in enumerate(attrs):
                out[r, c] = live[lv_row[a] * S + st, e] if a in lv_row else data.capacity[st] if a == 4 else data.day_weekday[data.tick_day[t]]
        return out
    for reverse in ((False, True) if wave else (False,)):
        be = CbEmuBackend(data, n, durations=260 if wave else 500, snapshot_resolution=10, max_snapshots=8, specialized=specialized, wave_decisions=wave, reverse=reverse)
        be.reset(transfer_times=draw_transfer_times(data, np.arange(n) + 3, be.layout.transfer_times_cap))
        obs = be.set_observation(attrs)
        S, cap = data.n_stations, be.layout.scope_cap
        dec, scope, met, done = be.step()
        checked = 0
        for budget in (0, 9):
            be.set_step_budget(budget)
            for i in range(1, 70 if wave else 120):
                for e in range(n):
                    valid = dec[e, 5] == 1 and not done[e]
                    if not valid:
                        assert not obs[e].any()
                        continue
                    nodes = scope[e, :, 0] if wave else np.arange(S)
                    want = expected(be, e, nodes, int(dec[e, 0]))
                    if not specialized:   # (generic build: the harness's query agrees as well)
                        assert np.array_equal(want, be.query(0, dec[:, 3:4], np.broadcast_to(np.where(nodes < 0, S + 7, nodes), (n, len(nodes))).copy(), attrs, len(attrs))[e, 0])
                    assert np.array_equal(obs[e], want), (topology, reverse, budget, i, e)
                    checked += 1
                a, na = be.random_policy(dec, scope, i)
                dec, scope, met, done = be.step(a, na)
        assert checked > 100


@pytest.mark.parametrize("specialized", [False, True])
def test_counting_rank_packed_and_general_forms(specialized):
    """cb::cbw_rank — the order every filter of the wave kernels' action scope sorts by: (value, station) descending (modes 0 / 2)
    or ascending (mode 1), as a rank per candidate.  Values of at most 20 bits take the one-word form (eight candidates per loop
    trip, positions >= n padded); anything larger or negative takes the general two-key loop.  Both against a plain sort, at the
    sizes around the loop's trip length and the two-per-lane register split, ties in the value included."""
    data = load_topology("city.180s")
    b = CbEmuBackend(data, n_envs=1, durations=60, snapshot_resolution=10, specialized=specialized, wave_decisions=2 if specialized else 1)
    rng = np.random.default_rng(11)
    for n in (1, 2, 7, 8, 9, 40, 63, 64, 65, 80, 127, 128):
        for hi in (3, 50, (1 << 20) - 1, 1 << 20, 1 << 30):
            for mode in (0, 1, 2):
                v = rng.integers(0, hi + 1, n).astype(np.int32)
                if hi == 1 << 30:
                    v[rng.integers(0, n)] = -5      # (no filter produces one; the general loop orders it like any int)
                key = rng.permutation(data.n_stations)[:n].astype(np.int32) if n <= data.n_stations else np.arange(n, dtype=np.int32)
                order = sorted(range(n), key=lambda i: (int(v[i]), int(key[i])), reverse=mode != 1)
                want = np.empty(n, np.int32)
                want[order] = np.arange(n)
                assert np.array_equal(b.rank(n, mode, v, key), want), (n, hi, mode)


@pytest.mark.parametrize("entries", [0, 2, 256])
@pytest.mark.parametrize("topology,kwargs,budget", [
    ("toy.5s_filters", dict(durations=700, snapshot_resolution=10, max_snapshots=7), 0),
    ("toy.3s_tight", dict(durations=900, snapshot_resolution=3, max_snapshots=9), 7),
    ("city.180s", dict(start_tick=1440, durations=70, snapshot_resolution=10, max_snapshots=6), 40),
])
def test_delivery_pool_staged_in_lds_by_the_wave_replay_step(monkeypatch, topology, kwargs, budget, entries):
    """mrx_k_cb_replay_wave reads the env's landing-tick buckets and the first K.pool_stage entries of its delivery ring out of an LDS
    copy (writes go to both).  The window at full size (what the product passes), at two entries (most of the ring falls outside: reads
    mix the copy and HBM, pushes land on either side, the ring wraps past the anchor) and off — same trajectories as the oracle."""
    from tests.cb_batch_check import run_bounded_vs_oracle
    monkeypatch.setenv("MRX_CB_AOS", "1")
    data = load_topology(topology)
    n = 4
    b = CbEmuBackend(data, n_envs=n, max_actions=1, specialized=True, wave_decisions=2, **kwargs)
    assert b.layout.env_major == 1
    b.set_pool_stage(entries)
    if budget:
        calls, unready = run_bounded_vs_oracle(b, data, kwargs, seeds=np.arange(n) + 31, budget=budget)
        assert unready > 0
    else:
        assert run_batch_vs_oracle(b, data, kwargs, seeds=np.arange(n) + 31, episodes=2) > 20
    handled, general = b.wave_counts()
    assert handled > 0 and general > 0


@pytest.mark.parametrize("overlap", [True, False])
@pytest.mark.parametrize("topology,kwargs,budget,period", [
    ("toy.5s_filters", dict(durations=700, snapshot_resolution=10, max_snapshots=7), 9, 2),
    ("toy.3s_tight", dict(durations=900, snapshot_resolution=3, max_snapshots=9), 7, 3),
    ("city.180s", dict(start_tick=1440, durations=70, snapshot_resolution=10, max_snapshots=6), 40, 2),
])
def test_replay_period_defers_envs_without_changing_trajectories(topology, kwargs, budget, period, overlap):
    """mrx_cb_set_replay_period: the general step runs on every n-th call only; in between an env that leaves its tick keeps the answer
    it was given (CbParams::stash, CFL_STASH) and reports "no decision yet".  Every env still follows the oracle decision by decision —
    the deferred answers are applied when the replay runs, later answers to the invalid rows are ignored."""
    from tests.cb_batch_check import run_bounded_vs_oracle
    data = load_topology(topology)
    n = 5
    b = CbEmuBackend(data, n_envs=n, max_actions=1, specialized=True, wave_decisions=2, **kwargs)
    b.set_replay_overlap(overlap)
    b.set_replay_period(period)
    calls, unready = run_bounded_vs_oracle(b, data, kwargs, seeds=np.arange(n) + 41, budget=budget)
    handled, general = b.wave_counts()
    assert unready > 0 and handled > 0 and general > 0
