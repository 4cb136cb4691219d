"""Parity tests proper: the HIP kernels (through the C ABI of include/maro_amd.h) against the
reference golden vectors and against the CPU oracle at batch scale.  Bit-exact (integer state;
the stochastic order generator is reproduced exactly, so no tolerance is needed)."""
import numpy as np
import pytest

from tests.golden_util import MATRIX_ATTRS, PORT_ATTRS, VESSEL_ATTRS, golden_cases, joint_golden_cases

pytestmark = pytest.mark.gpu


def _make(topo, kwargs):
    from tests.backend_adapter import SingleEnvAdapter
    from tests.gpu_backend import GpuBackend
    b = GpuBackend(topo, n_envs=1, start_tick=kwargs.get("start_tick", 0), durations=kwargs["durations"], snapshot_resolution=kwargs.get("snapshot_resolution", 1),
                   max_snapshots=kwargs.get("max_snapshots"), max_actions=2)
    return SingleEnvAdapter(b)


@pytest.mark.parametrize("name", golden_cases())
def test_hip_engine_reproduces_reference(name):
    from tests.test_oracle_golden import replay_case
    replay_case(_make, name)


@pytest.mark.parametrize("name", joint_golden_cases())
def test_hip_engine_joint_decision_modes(name):
    """DecisionMode.Joint / JointWithSequentialAction through mrx_cim_step_joint."""
    from tests.gpu_backend import GpuBackend
    from tests.golden_util import replay_joint_case
    from tests.test_emu_joint import JointAdapter

    def make(topo, kwargs, mode):
        b = GpuBackend(topo, n_envs=3, start_tick=kwargs.get("start_tick", 0), durations=kwargs["durations"], snapshot_resolution=kwargs.get("snapshot_resolution", 1),
                       max_snapshots=kwargs.get("max_snapshots"), max_actions=topo.n_vessels, decision_mode=mode)
        return JointAdapter(b, env=2)
    replay_joint_case(make, name)


def test_extension_is_loaded_not_a_fallback():
    import maro_amd._lib as L
    lib = L.load()
    assert b"gfx950" in lib.mrx_version()
    with open("/proc/self/maps") as fp:
        assert "libmaro_amd.so" in fp.read()


@pytest.mark.parametrize("topology,n_envs,durations", [("toy.4p_ssdd_l0.0", 256, 120), ("global_trade.22p_l0.8", 96, 70),
                                                       ("toy.5p_ssddd_l0.6", 64, 100)])
def test_batch_matches_oracle_per_seed(topology, n_envs, durations):
    """BASELINE config 2 shape: many envs, env e seeded e (set_seed(e)+reset), counter-based random legal
    actions; every env/step payload and the final snapshot tensors compared with the CPU oracle."""
    import torch

    from maro_amd.cim.engine import CimBatchEngine
    from oracle.cim_oracle import CimOracle, hash_policy_action

    seeds = np.arange(n_envs, dtype=np.int64) * 7 + 3
    eng = CimBatchEngine(topology, n_envs, durations=durations, max_actions=1, seeds=seeds)
    oracles = []
    for s in seeds:
        o = CimOracle(topology, durations=durations)
        o.set_seed(int(s))
        o.reset(keep_seed=True)
        oracles.append(o)
    ostate = [o.step(None) for o in oracles]
    dec, met, done = (x.cpu().numpy() for x in eng.step())
    step = 0
    while True:
        acts = np.zeros((n_envs, 1, 4), np.int32)
        nact = np.zeros(n_envs, np.int32)
        all_done = True
        for e, (om, od, odone) in enumerate(ostate):
            assert bool(done[e]) == odone, (e, step)
            assert np.array_equal(met[e], om), (e, step, met[e], om)
            if not odone:
                all_done = False
                assert np.array_equal(dec[e], od), (e, step, dec[e], od)
                acts[e, 0] = hash_policy_action(int(seeds[e]), step, od)
                nact[e] = 1
        if all_done:
            break
        ostate = [o.step([tuple(acts[e, 0])]) if not ostate[e][2] else ostate[e] for e, o in enumerate(oracles)]
        dec, met, done = (x.cpu().numpy() for x in eng.step(acts, nact, mask=(1 - done).astype(np.uint8)))
        step += 1
    assert int(eng.status.cpu().abs().sum()) == 0
    ticks = np.arange(0, durations, dtype=np.int32)
    for node, attrs in (("ports", PORT_ATTRS), ("vessels", VESSEL_ATTRS), ("matrices", MATRIX_ATTRS)):
        n_nodes = {"ports": eng.topo.n_ports, "vessels": eng.topo.n_vessels, "matrices": 1}[node]
        got = eng.query(node, ticks, np.arange(n_nodes, dtype=np.int32), attrs).cpu().numpy()
        for e in range(0, n_envs, max(1, n_envs // 16)):
            exp = oracles[e].query(node, ticks, [], attrs)
            assert np.array_equal(got[e].reshape(-1), exp), (node, e)


def test_every_packaged_topology_matches_oracle():
    """All 36 shipped topologies, 4 envs each (different seeds), 60 ticks, counter-based random agent."""
    from maro_amd.cim.engine import CimBatchEngine
    from maro_amd.cim.topology import available_topologies
    from oracle.cim_oracle import CimOracle, hash_policy_action

    for topology in available_topologies():
        n, dur = 4, 60
        seeds = np.array([4096, 1, 77, 123456], np.int64)
        eng = CimBatchEngine(topology, n, durations=dur, max_actions=1, seeds=seeds)
        oracles = []
        for s in seeds:
            o = CimOracle(topology, durations=dur)
            o.set_seed(int(s))
            o.reset(keep_seed=True)
            oracles.append(o)
        ost = [o.step(None) for o in oracles]
        dec, met, done = (x.cpu().numpy() for x in eng.step())
        step = 0
        while not all(s[2] for s in ost):
            acts = np.zeros((n, 1, 4), np.int32)
            nact = np.zeros(n, np.int32)
            for e, (om, od, odone) in enumerate(ost):
                assert bool(done[e]) == odone and np.array_equal(met[e], om), (topology, e, step)
                if not odone:
                    assert np.array_equal(dec[e], od), (topology, e, step, dec[e], od)
                    acts[e, 0] = hash_policy_action(int(seeds[e]), step, od)
                    nact[e] = 1
            ost = [o.step([tuple(acts[e, 0])]) if not ost[e][2] else ost[e] for e, o in enumerate(oracles)]
            dec, met, done = (x.cpu().numpy() for x in eng.step(acts, nact, mask=(1 - done).astype(np.uint8)))
            step += 1
        assert int(eng.status.cpu().abs().sum()) == 0, topology
        got = eng.query("matrices", np.arange(dur, dtype=np.int32), np.zeros(1, np.int32), MATRIX_ATTRS).cpu().numpy()
        for e in range(n):
            assert np.array_equal(got[e].reshape(-1), oracles[e].query("matrices", list(range(dur)), [], MATRIX_ATTRS)), (topology, e)


def test_full_size_bench_workload_properties():
    """BASELINE configs[2] at full size: 16384 envs of global_trade.22p_l0.8, whole 1120-tick episodes with the
    device-side random agent.  Checked: (1) container conservation in every env after every 200 steps,
    (2) envs sharing a seed stay bit-identical, (3) 24 sampled envs end with exactly the oracle's metrics,
    (4) no env raised a status flag, every env finishes at tick 1119."""
    import torch

    from maro_amd.cim.engine import CimBatchEngine
    from oracle.cim_oracle import CimOracle

    n, dur, topology = 16384, 1120, "global_trade.22p_l0.8"
    seeds = torch.arange(n, dtype=torch.int64) + 1
    seeds[n // 2:] = seeds[: n // 2]  # second half replays the first half's seeds
    eng = CimBatchEngine(topology, n, durations=dur, max_snapshots=2, max_actions=1, seeds=seeds)
    lay, P, V = eng.layout, eng.topo.n_ports, eng.topo.n_vessels
    actions = torch.zeros((n, 1, 4), dtype=torch.int32, device="cuda")
    nact = torch.zeros((n,), dtype=torch.int32, device="cuda")

    def containers():
        live = eng.live.to(torch.int64)
        ports = live[:, lay.frame_off_ports: lay.frame_off_ports + 12 * P].view(n, 12, P)
        ves = live[:, lay.frame_off_vessels: lay.frame_off_vessels + 3 * V].view(n, 3, V)
        return ports[:, 1:5].sum(dim=(1, 2)) + ves[:, 1:3].sum(dim=(1, 2))   # empty, full, on_shipper, on_consignee | empty, full

    total0 = containers()
    assert int(total0.min()) == int(total0.max())
    eng.step()
    i = 1
    while True:
        eng.random_policy(i - 1, actions, nact)          # step counter i-1 == the oracle rollout's decision index
        _, _, done = eng.step(actions, nact, mask=(eng.done == 0).to(torch.uint8))
        if i % 200 == 0:
            assert torch.equal(containers(), total0), f"containers not conserved after {i} steps"
            if bool(done.all()):
                break
        i += 1
        assert i < 6000
    assert int((eng.status != 0).sum()) == 0
    assert int(eng.ticks.min()) == dur - 1 == int(eng.ticks.max())
    met = eng.metrics.cpu().numpy()
    assert np.array_equal(met[: n // 2], met[n // 2:])
    assert torch.equal(eng.live[: n // 2], eng.live[n // 2:])
    for e in list(range(0, 16)) + [777, 4095, 5000, 8191, 8192, 12345, 16000, 16383]:
        o = CimOracle(topology, durations=dur)
        o.set_seed(int(seeds[e]))
        o.reset(keep_seed=True)
        _, _, om = o.rollout(int(seeds[e]))
        assert np.array_equal(met[e], om), (e, met[e], om)


@pytest.mark.parametrize("topology,n", [("global_trade.22p_l0.8", 130), ("toy.5p_ssddd_l0.5", 70)])
def test_fused_observation_matches_snapshot_slices(topology, n):
    from maro_amd.cim.topology import load_topology
    from tests.gpu_backend import GpuBackend
    from tests.test_emu_observation import check_fused_observation
    b = GpuBackend(load_topology(topology), n_envs=n, durations=80, max_actions=1)
    check_fused_observation(b, seeds=np.arange(n) + 5)


def test_config2_4096_envs_bit_exact_per_seed():
    """BASELINE.json configs[1]: CIM toy.4p_ssdd_l0.0, 4096 parallel envs on one GPU, env e seeded e (seed 4096 for e = 0,
    the topology default), device random policy; 96 sampled envs replayed through the oracle: every payload, metric, action
    and the final snapshot tensors bit-exact."""
    import torch

    from maro_amd.cim.engine import CimBatchEngine
    from oracle.cim_oracle import CimOracle, hash_policy_action

    n, dur, topology = 4096, 200, "toy.4p_ssdd_l0.0"
    seeds = np.arange(n, dtype=np.int64)
    seeds[0] = 4096
    eng = CimBatchEngine(topology, n, durations=dur, max_actions=1, seeds=seeds)
    sample = sorted(set([0, 1, 63, 64, 65, n - 1] + list(np.random.RandomState(0).choice(n, 90, replace=False))))
    oracles = {}
    for e in sample:
        o = CimOracle(topology, durations=dur)
        o.set_seed(int(seeds[e]))
        o.reset(keep_seed=True)
        oracles[e] = o
    ost = {e: o.step(None) for e, o in oracles.items()}
    actions = torch.zeros((n, 1, 4), dtype=torch.int32, device=eng.device)
    nact = torch.zeros(n, dtype=torch.int32, device=eng.device)
    dec, met, done = (x.cpu().numpy() for x in eng.step())
    step = 0
    while not done.all():
        eng.random_policy(step, actions, nact)
        a = actions.cpu().numpy()
        for e, o in oracles.items():
            om, od, odone = ost[e]
            assert bool(done[e]) == odone and np.array_equal(met[e], om), (e, step)
            if not odone:
                assert np.array_equal(dec[e], od), (e, step, dec[e], od)
                want = hash_policy_action(int(seeds[e]), step, od)
                assert tuple(a[e, 0]) == want, (e, step, a[e, 0], want)
                ost[e] = o.step([want])
        dec, met, done = (x.cpu().numpy() for x in eng.step(actions, nact))
        step += 1
        assert step < 1000
    assert int(eng.status.cpu().abs().sum()) == 0
    for e, o in oracles.items():
        assert ost[e][2] and np.array_equal(met[e], ost[e][0]), e
    ticks = np.arange(dur, dtype=np.int32)
    for node, attrs in (("ports", PORT_ATTRS), ("vessels", VESSEL_ATTRS), ("matrices", MATRIX_ATTRS)):
        n_nodes = {"ports": eng.topo.n_ports, "vessels": eng.topo.n_vessels, "matrices": 1}[node]
        got = eng.query(node, ticks, np.arange(n_nodes, dtype=np.int32), attrs).cpu().numpy()
        for e in sample[::6]:
            assert np.array_equal(got[e].reshape(-1), oracles[e].query(node, ticks, [], attrs)), (node, e)
