"""citi_bike on the MI355X through the C ABI: reference goldens, batches vs the oracle, full-size properties."""
import numpy as np
import pytest

from maro_amd.citi_bike.data import load_topology
from tests.cb_backend_adapter import CbBackendEnv
from tests.cb_batch_check import run_batch_vs_oracle
from tests.test_citi_bike_oracle import CASES, replay_citi_bike

pytestmark = pytest.mark.gpu


def make(data, kw, tt, n_envs=70):
    from tests.cb_gpu_backend import CbGpuBackend
    b = CbGpuBackend(data, n_envs=n_envs, max_actions=1, **kw)
    b.reset(transfer_times=[tt[: b.layout.transfer_times_cap]] * n_envs)
    return CbBackendEnv(b, env=n_envs - 1)


@pytest.mark.parametrize("case", CASES)
def test_hip_engine_reproduces_reference(case):
    replay_citi_bike(make, case)


def _joint_cases():
    from tests.test_citi_bike_joint_oracle import JOINT_CASES
    return JOINT_CASES


@pytest.mark.parametrize("case", _joint_cases())
def test_hip_engine_reproduces_reference_joint_modes(case):
    """mrx_cb_step_joint (Joint / JointWithSequentialAction) against the reference's vectors."""
    from tests.cb_gpu_backend import CbGpuBackend
    from tests.test_citi_bike_joint_oracle import replay_citi_bike_joint

    def make_joint(data, kw, tt, mode, n_envs=70):
        b = CbGpuBackend(data, n_envs=n_envs, max_actions=1, decision_mode=mode, **kw)
        b.reset(transfer_times=[tt[: b.layout.transfer_times_cap]] * n_envs)
        return CbBackendEnv(b, env=n_envs - 1)
    replay_citi_bike_joint(make_joint, case)


@pytest.mark.parametrize("topology,kwargs,n", [
    ("toy.3s_4t", dict(durations=1440, snapshot_resolution=10), 200),
    ("toy.3s_tight", dict(durations=1100, snapshot_resolution=7, max_snapshots=9), 130),
])
def test_batch_matches_oracle(topology, kwargs, n):
    from tests.cb_gpu_backend import CbGpuBackend
    data = load_topology(topology)
    b = CbGpuBackend(data, n_envs=n, max_actions=1, **kwargs)
    run_batch_vs_oracle(b, data, kwargs, seeds=np.arange(n) + 11, episodes=2, check_envs=[0, 1, 63, 64, 65, n - 1])


@pytest.mark.parametrize("S,seed,n", [(180, 1, 130), (333, 2, 70)])
def test_city_sized_data(S, seed, n):
    """City-shaped synthetic data (maro_amd/citi_bike/synthetic.py): hundreds of stations, sparse neighbour lists, the
    default three-filter chain, the HBM-frame station loops."""
    from tests.cb_gpu_backend import CbGpuBackend
    from tests.fuzz_citi_bike import run_city_case
    assert run_city_case(seed, S, backend=CbGpuBackend, n_envs=n) > 100


def test_packaged_city_topology_day():
    """The packaged city.180s topology, one day at resolution 10: three envs of a 200-env batch against the oracle."""
    from tests.cb_gpu_backend import CbGpuBackend
    data = load_topology("city.180s")
    kw = dict(durations=1440, snapshot_resolution=10)
    b = CbGpuBackend(data, n_envs=200, max_actions=1, **kw)
    run_batch_vs_oracle(b, data, kw, seeds=np.arange(200) + 5, episodes=1, check_envs=[0, 77, 199])


@pytest.mark.parametrize("budget,lanes", [(5, 64), (24, 16), (3, 1)])
def test_bounded_steps_and_lanes_per_wave_do_not_change_trajectories(budget, lanes):
    """mrx_cb_set_step_budget / mrx_cb_set_lanes_per_wave only regroup the work: every env still follows the oracle."""
    from tests.cb_batch_check import run_bounded_vs_oracle
    from tests.cb_gpu_backend import CbGpuBackend
    data = load_topology("toy.5s_6t")
    kw = dict(durations=700, snapshot_resolution=5)
    b = CbGpuBackend(data, n_envs=150, max_actions=1, **kw)
    b.eng.set_lanes_per_wave(lanes)
    calls, unready = run_bounded_vs_oracle(b, data, kw, seeds=np.arange(150) + 3, budget=budget, check_envs=[0, 1, 63, 64, 149])
    assert unready > 0


@pytest.mark.parametrize("case_seed,mode,budget", [(7, 2, 0), (11, 1, 9), (23, 2, 5)])
def test_joint_modes_on_random_data(case_seed, mode, budget):
    """Joint modes (+ bounded steps) on random data sets, 100 envs with per-env random numbers of answered events, vs the oracle."""
    from tests.cb_batch_check import run_joint_vs_oracle
    from tests.cb_gpu_backend import CbGpuBackend
    from tests.fuzz_citi_bike import random_data
    rng = np.random.RandomState(case_seed)
    data = random_data(rng)
    kw = dict(durations=int(rng.choice([150, 400])), snapshot_resolution=int(rng.choice([1, 4, 10])))
    b = CbGpuBackend(data, n_envs=100, max_actions=1, decision_mode=mode, **kw)
    calls, events = run_joint_vs_oracle(b, data, kw, seeds=np.arange(100) + case_seed, mode=mode, budget=budget, check_envs=[0, 1, 63, 64, 99])
    assert events > 100


def test_full_size_batch_properties():
    """BASELINE config 4 size on one GPU (4096 envs, a month-long... one day here): conservation laws that hold for any
    trajectory: trips = fulfilled + shortage; bikes are conserved up to in-flight / lost ones; identical seeds and
    actions give identical envs."""
    import torch
    from maro_amd.citi_bike.engine import CitiBikeBatchEngine
    n = 4096
    seeds = np.arange(n) % 1024  # envs e and e + 1024k share a seed
    eng = CitiBikeBatchEngine("toy.3s_4t", n, durations=1440, snapshot_resolution=10, seeds=seeds)
    a = torch.zeros((n, 1, 3), dtype=torch.int32, device=eng.device)
    na = torch.zeros(n, dtype=torch.int32, device=eng.device)
    eng.step()
    step = 0
    while not bool(eng.done.all()):
        step += 1
        # policy keyed by step and env: fold the env index so that seed-sharing envs act alike
        eng.random_policy(step, a, na)
        a4 = a.view(4, 1024, 3)
        a4[1:] = a4[0]
        na.view(4, 1024)[1:] = na.view(4, 1024)[0]
        eng.step(a, na)
        assert step < 400
    torch.cuda.synchronize()
    assert int(eng.status.abs().sum()) == 0
    m = eng.metrics.cpu().numpy()
    assert (m.reshape(4, 1024, 3) == m[:1024]).all()
    S = eng.data.n_stations
    ntrips = int((eng.data.trip_tick < 1440).sum())
    assert (m[:, 0] == ntrips).all()
    fis = np.arange(144, dtype=np.int32)
    q = eng.query("stations", fis, np.arange(S, dtype=np.int32), ["shortage", "fulfillment", "trip_requirement"]).cpu().numpy()
    assert (q[..., 0] + q[..., 1] == q[..., 2]).all()
    assert (q[..., 2].sum(axis=(1, 2)) == ntrips).all() and (q[..., 0].sum(axis=(1, 2)) == m[:, 1]).all()
    adj = eng.query("matrices", fis[-1:], np.zeros(1, np.int32), ["trips_adj"]).cpu().numpy()
    assert (adj.sum(axis=(1, 2, 3)) == ntrips).all()


# ---- the reference's own size: city.800s = 800 stations, ny filter chain 80 -> 40 -> 20 over 10 windows, a month of ticks
def test_city800_batch_matches_oracle():
    """A 300-env batch of city.800s (wave-cooperative decision step + general kernel, as mrx_cb_step runs them at this size), per-env
    actions and transfer times, three envs replayed on the oracle: every decision, scope, metric and the snapshot history."""
    from tests.cb_gpu_backend import CbGpuBackend
    data = load_topology("city.800s")
    kw = dict(start_tick=1440, durations=130, snapshot_resolution=10, max_snapshots=6)
    b = CbGpuBackend(data, n_envs=300, max_actions=1, **kw)
    assert b.eng.set_wave_decisions(0)   # automatic = on at this size
    steps = run_batch_vs_oracle(b, data, kw, seeds=np.arange(300) + 17, episodes=1, check_envs=[0, 151, 299])
    assert steps > 1000


@pytest.mark.parametrize("overlap,period", [(True, 1), (False, 1), (False, 3)])
def test_city800_batch_matches_oracle_specialised(overlap, period):
    """The same batch on the plan-specialised LDS-frame kernels: decision step AND general step one env per wave
    (mrx_k_cb_step_wave + mrx_k_cb_replay_wave), with a step budget on top — what bench.py --topology city.800s runs.  overlap:
    the two kernels side by side on disjoint envs (mrx_k_cb_classify first; mrx_cb_set_replay_overlap) or in sequence; period: mrx_cb_set_replay_period."""
    from tests.cb_batch_check import run_bounded_vs_oracle
    from tests.cb_gpu_backend import CbGpuBackend
    data = load_topology("city.800s")
    kw = dict(start_tick=1440, durations=130, snapshot_resolution=10, max_snapshots=6)
    b = CbGpuBackend(data, n_envs=300, max_actions=1, specialize=True, **kw)
    assert b.eng.specialized and b.eng.set_wave_decisions(0)
    b.eng.set_replay_overlap(overlap)
    b.eng.set_replay_period(period, period - 1)   # > 1: the replay kernel on every n-th call, deferred envs (stashed answers) in between (mrx_cb_set_replay_period)
    calls, unready = run_bounded_vs_oracle(b, data, kw, seeds=np.arange(300) + 17, budget=96, check_envs=[0, 151, 299])
    assert calls > 1000 and unready > 0


def test_replay_overlap_does_not_change_a_single_output():
    """mrx_cb_set_replay_overlap on / off on two engines of the same city-size batch, under a user stream and a tight step budget (many
    envs in the replay kernel at once): every output of every step and the final state words are equal, and the caller's stream sees
    the step as one ordered operation (the next launch on it reads both kernels' outputs)."""
    import torch

    from maro_amd.citi_bike.engine import CitiBikeBatchEngine
    n = 512
    engs = [CitiBikeBatchEngine("city.180s", n, durations=400, snapshot_resolution=10, max_snapshots=8, seeds=np.arange(n) + 5, specialize=True) for _ in range(2)]
    assert all(e.specialized and e.set_wave_decisions(0) for e in engs)
    engs[0].set_replay_overlap(True)
    engs[1].set_replay_overlap(False)
    user = torch.cuda.Stream()
    engs[0].use_stream(user)
    for e in engs:
        e.set_step_budget(5)
    acts = [torch.zeros((n, 1, 3), dtype=torch.int32, device="cuda") for _ in engs]
    nact = [torch.zeros(n, dtype=torch.int32, device="cuda") for _ in engs]
    sums = [torch.zeros(4, dtype=torch.int64, device="cuda") for _ in engs]
    masks = [(torch.arange(n, device="cuda") % 3 != r).to(torch.uint8) for r in range(3)]
    torch.cuda.synchronize()
    outs = [e.step() for e in engs]
    for i in range(600):
        for e, a, k in zip(engs, acts, nact):
            e.random_policy(i, a, k)
        mask = masks[i % 3] if i % 7 == 0 else None     # (a masked step now and then: masked envs belong to neither kernel)
        outs = [e.step(a, k, mask) for e, a, k in zip(engs, acts, nact)]
        with torch.cuda.stream(user):   # stream-ordered consumer right behind the step: no host sync in between
            sums[0] += torch.stack([o.to(torch.int64).sum() for o in outs[0]])
        sums[1] += torch.stack([o.to(torch.int64).sum() for o in outs[1]])
        if i % 50 == 0:
            torch.cuda.synchronize()
            for x, y in zip(*outs):
                assert torch.equal(x, y), i
    torch.cuda.synchronize()
    assert torch.equal(sums[0], sums[1])
    for name in ("hdr", "live", "ring", "ring_fi"):
        assert torch.equal(getattr(engs[0], name), getattr(engs[1], name)), name


@pytest.mark.parametrize("topology,kwargs", [("toy.5s_filters", dict(durations=1200, snapshot_resolution=10)),
                                             ("toy.3s_tight", dict(durations=1000, snapshot_resolution=5, max_snapshots=11))])
def test_wave_cooperative_decisions_on_small_topologies(topology, kwargs):
    """Forced on (automatic only switches it on from 96 stations): same trajectories as the oracle."""
    from tests.cb_gpu_backend import CbGpuBackend
    data = load_topology(topology)
    b = CbGpuBackend(data, n_envs=130, max_actions=1, **kwargs)
    assert b.eng.set_wave_decisions(1)
    run_batch_vs_oracle(b, data, kwargs, seeds=np.arange(130) + 3, episodes=2, check_envs=[0, 63, 64, 129])


def test_city800_full_month_properties():
    """The whole 44 640-tick horizon of city.800s (1.5 M trips) on a 128-env batch, on properties only.  Joint decision mode: a
    step call reports every pending decision of a decision tick at once (~200 per env at this size) and the next call finishes
    them; the first reported event of every tick is answered (half the scope to the nearest candidate).  Properties: every RequireBike is counted (trips
    metric = trips of the month), shortage + fulfilment = requirement in every frame of the ring, envs that share a seed and act
    alike stay identical, bikes only leave the system through deliveries still in flight, no status bit is raised."""
    import dataclasses

    import torch
    from maro_amd.citi_bike.engine import CitiBikeBatchEngine
    n, dur = 128, 44640
    seeds = np.arange(n) % 64   # envs e and e + 64 share seed and actions
    # one decision tick per DAY (31 of them, ~200 decision events each) instead of one every 20 ticks: the month's 3 M trip /
    # return events, 744 frames through a 24-slot ring and the month-long shared tables are what this test is about — at the
    # topology's own resolution the 450 000 decision events of a month would dominate it (the batch-vs-oracle test above and the
    # goldens cover those at resolution 20)
    data = dataclasses.replace(load_topology("city.800s"), resolution=1440)
    eng = CitiBikeBatchEngine(data, n, durations=dur, snapshot_resolution=60, max_snapshots=24, seeds=seeds, decision_mode=1,
                              transfer_times_cap=4096)
    S = eng.data.n_stations
    a = torch.zeros((n, S, 1, 3), dtype=torch.int32, device=eng.device)
    na = torch.zeros((n, S), dtype=torch.int32, device=eng.device)
    nans = torch.zeros(n, dtype=torch.int32, device=eng.device)
    dec, scope, met, done = eng.step_joint()
    calls = 0
    while not bool(done.all()):
        calls += 1
        # answer the FIRST reported event of every env: move half of what the scope allows between the station and its first candidate
        d0, s0 = dec[:, 0], scope[:, 0]
        valid = (d0[:, 5] == 1) & (d0[:, 4] >= 2)
        k = (d0[:, 4] - 1).clamp(min=0).long()
        self_row = s0[torch.arange(n, device=eng.device), k]
        num = torch.minimum(self_row[:, 1], s0[:, 0, 1]).clamp(min=0) // 2
        supply = d0[:, 2] == 0
        a[:, 0, 0, 0] = torch.where(supply, self_row[:, 0], s0[:, 0, 0])
        a[:, 0, 0, 1] = torch.where(supply, s0[:, 0, 0], self_row[:, 0])
        a[:, 0, 0, 2] = num
        na[:, 0] = valid.to(torch.int32)
        nans[:] = valid.to(torch.int32)
        dec, scope, met, done = eng.step_joint(a, na, nans)
        assert calls < 200
    torch.cuda.synchronize()
    assert calls >= dur // eng.data.resolution - 1
    assert int(eng.status.abs().sum()) == 0
    m = eng.metrics.cpu().numpy()
    assert (m[:64] == m[64:]).all() and (m[:, 2] > 0).all()
    assert (m[:, 0] == int((eng.data.trip_tick < dur).sum())).all()
    fis = np.arange(dur // 60 - 24, dur // 60, dtype=np.int32)
    q = eng.query("stations", fis, np.arange(S, dtype=np.int32), ["shortage", "fulfillment", "trip_requirement", "bikes"]).cpu().numpy()
    assert (q[..., 0] + q[..., 1] == q[..., 2]).all() and q[..., 2].sum() > 0
    total0 = int(eng.data.init_bikes.sum())
    assert (q[:, -1, :, 3].sum(-1) <= total0).all() and (q[:, -1, :, 3].sum(-1) > 0.5 * total0).all()   # the rest is on trips / in delivery


@pytest.mark.parametrize("topology,specialize", [("toy.5s_6t", False), ("toy.5s_6t", True), ("city.180s", True)])
def test_fused_observation_equals_the_query(topology, specialize):
    """mrx_cb_set_observation: the stations slice of every env's new decision, written by the step kernels themselves, equals the
    mrx_cb_query launch it replaces at every step — every station's row on the one-env-per-lane kernels (generic and LDS-frame),
    the rows of the action scope's stations on the wave-cooperative kernels (city.180s), with and without bounded steps (rows of envs
    without a decision are zeros)."""
    import torch
    from maro_amd.citi_bike.engine import CitiBikeBatchEngine
    attrs = ["bikes", "shortage", "trip_requirement", "fulfillment", "capacity", "extra_cost", "min_bikes", "weekday"]
    n = 200
    eng = CitiBikeBatchEngine(topology, n, durations=700, snapshot_resolution=10, max_snapshots=16, specialize=specialize, seeds=np.arange(n) + 5)
    S, wave = eng.data.n_stations, eng.set_wave_decisions(0)
    obs = eng.set_observation(attrs)
    assert obs.shape == (n, eng.layout.scope_cap if wave else S, len(attrs))
    stations = torch.arange(S, dtype=torch.int32, device=eng.device)
    a = torch.zeros((n, 1, 3), dtype=torch.int32, device=eng.device)
    na = torch.zeros(n, dtype=torch.int32, device=eng.device)
    eng.step()
    nonzero = 0
    for budget in (0, 7):
        eng.set_step_budget(budget)
        for i in range(1, 160):
            nodes = eng.scope[:, :, 0].contiguous() if wave else stations      # (-1 padding reads as zeros: query semantics)
            want = eng.query("stations", eng.decisions[:, 3:4], nodes, attrs)[:, 0]
            valid = (eng.decisions[:, 5] == 1) & (eng.done == 0)
            assert torch.equal(obs[valid], want[valid]), (budget, i)
            assert not bool(obs[~valid].any())
            nonzero += int(valid.sum())
            eng.random_policy(i, a, na)
            eng.step(a, na)
    assert nonzero > 10000 and int(eng.status.abs().sum()) == 0
    assert eng.set_observation(()) is None


def test_fused_observation_pins_the_step_path():
    """the fused observation's buffer is sized for ONE row layout (every station on the lane path, the action scope's
    stations on the wave path).  `set_observation` must not touch the wave mode the caller chose (it used to reset it to automatic),
    and while an observation is configured a switch of the step path is refused instead of letting the other kernel write
    S rows per env into a scope_cap-row buffer."""
    import torch
    from maro_amd.citi_bike.engine import CitiBikeBatchEngine
    attrs = ["bikes", "capacity"]
    n = 64
    # a city-size plan, specialised: automatic = wave path.  The caller switches the wave kernels OFF, then asks for an observation
    eng = CitiBikeBatchEngine("city.180s", n, durations=300, snapshot_resolution=10, max_snapshots=8, specialize=True, seeds=np.arange(n) + 1)
    assert eng.set_wave_decisions(0) is True
    assert eng.set_wave_decisions(-1) is False
    obs = eng.set_observation(attrs)
    assert obs.shape[1] == eng.data.n_stations                      # lane path: every station
    assert eng.set_wave_decisions(-1) is False                      # ... and the caller's choice is still in force (no-op call)
    with pytest.raises(RuntimeError, match="fused observation"):
        eng.set_wave_decisions(1)                                   # would make the wave kernels write scope rows into this buffer
    with pytest.raises(RuntimeError, match="fused observation"):
        eng.set_wave_decisions(0)                                   # automatic = on at 180 stations: refused as well
    a = torch.zeros((n, 1, 3), dtype=torch.int32, device=eng.device)
    na = torch.zeros(n, dtype=torch.int32, device=eng.device)
    stations = torch.arange(eng.data.n_stations, dtype=torch.int32, device=eng.device)
    eng.step()
    for i in range(1, 40):
        want = eng.query("stations", eng.decisions[:, 3:4], stations, attrs)[:, 0]
        valid = (eng.decisions[:, 5] == 1) & (eng.done == 0)
        assert torch.equal(obs[valid], want[valid]), i
        eng.random_policy(i, a, na)
        eng.step(a, na)
    assert eng.set_observation(()) is None
    assert eng.set_wave_decisions(1) is True                        # with the observation off the switch is allowed again
    obs2 = eng.set_observation(attrs)
    assert obs2.shape[1] == eng.layout.scope_cap
    assert int(eng.status.abs().sum()) == 0
