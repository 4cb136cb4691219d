"""The plan-specialised step kernels (mrx_cim_load_step_kernels) on an MI355X: same device source with the plan's dimensions
as compile-time constants, so every parity check the generic kernels pass must pass unchanged."""
import ctypes

import numpy as np
import pytest

from tests.golden_util import golden_cases, joint_golden_cases

pytestmark = pytest.mark.gpu

SUBSET = [c for i, c in enumerate(golden_cases()) if i % 3 == 0]


class _Spec:
    """GpuBackend with specialize=True (and a check that the specialised module really is in use)."""

    def __new__(cls, *a, **kw):
        import os
        from tests.gpu_backend import GpuBackend
        old = os.environ.get("MARO_AMD_SPECIALIZE")
        os.environ["MARO_AMD_SPECIALIZE"] = "1"
        try:
            b = GpuBackend(*a, **kw)
        finally:
            if old is None:
                del os.environ["MARO_AMD_SPECIALIZE"]
            else:
                os.environ["MARO_AMD_SPECIALIZE"] = old
        assert b.eng.specialized
        return b


@pytest.mark.parametrize("name", SUBSET)
def test_specialized_kernels_reproduce_reference(name):
    from tests.backend_adapter import SingleEnvAdapter
    from tests.test_oracle_golden import replay_case

    def make(topo, kwargs):
        return SingleEnvAdapter(_Spec(topo, n_envs=1, start_tick=kwargs.get("start_tick", 0), durations=kwargs["durations"], snapshot_resolution=kwargs.get("snapshot_resolution", 1),
                                      max_snapshots=kwargs.get("max_snapshots"), max_actions=2))
    replay_case(make, name)


@pytest.mark.parametrize("name", joint_golden_cases()[:2])
def test_specialized_kernels_joint_modes(name):
    from tests.golden_util import replay_joint_case
    from tests.test_emu_joint import JointAdapter

    def make(topo, kwargs, mode):
        return JointAdapter(_Spec(topo, n_envs=3, start_tick=kwargs.get("start_tick", 0), durations=kwargs["durations"], snapshot_resolution=kwargs.get("snapshot_resolution", 1),
                                  max_snapshots=kwargs.get("max_snapshots"), max_actions=topo.n_vessels, decision_mode=mode), env=2)
    replay_joint_case(make, name)


@pytest.mark.parametrize("order_table", [0, -1])
def test_specialized_equals_generic_batch_with_fused_observation(order_table):
    """700 envs of global_trade.22p, device policy keyed on the decision: decisions, metrics, fused observations and the final
    workspace state of a specialised engine equal the generic engine's, bit for bit."""
    import torch

    from maro_amd.cim.engine import CimBatchEngine
    n, topo = 700, "global_trade.22p_l0.8"
    seeds = torch.arange(n, dtype=torch.int64) * 5 + 2
    engs = [CimBatchEngine(topo, n, durations=90, seeds=seeds, max_snapshots=5, order_table=order_table, specialize=s) for s in (False, True)]
    assert not engs[0].specialized and engs[1].specialized
    obs = [e.set_observation(["empty", "full", "shortage", "transfer_cost"], ["empty", "remaining_space"]) for e in engs]
    acts = [torch.zeros((n, 1, 4), dtype=torch.int32, device="cuda") for _ in engs]
    nact = [torch.zeros(n, dtype=torch.int32, device="cuda") for _ in engs]
    for e in engs:
        e.step()
    for _ in range(300):
        for e, a, k in zip(engs, acts, nact):
            e.random_policy(-1, a, k)
            e.step(a, k)
        torch.cuda.synchronize()
        assert torch.equal(engs[0].decisions, engs[1].decisions) and torch.equal(engs[0].metrics, engs[1].metrics)
        assert torch.equal(obs[0][0], obs[1][0]) and torch.equal(obs[0][1], obs[1][1])
    assert bool(engs[0].done.all()) and torch.equal(engs[0].done, engs[1].done)
    assert torch.equal(engs[0].live, engs[1].live) and torch.equal(engs[0].ring, engs[1].ring) and int(engs[1].status.max()) == 0


def test_changing_the_observation_reloads_the_specialized_kernels():
    """The fused observation's configuration is compiled in: set_observation with other attributes must not leave kernels built
    for the previous ones in place (the C side drops them; the Python engine loads the right ones)."""
    import torch

    from maro_amd.cim.engine import CimBatchEngine
    n, topo = 64, "toy.5p_ssddd_l0.5"
    eng = CimBatchEngine(topo, n, durations=60, specialize=True)
    ref = CimBatchEngine(topo, n, durations=60, specialize=False)
    acts, nact = torch.zeros((n, 1, 4), dtype=torch.int32, device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda")
    for attrs in ((["empty", "shortage"], ["remaining_space"]), (["full", "booking", "transfer_cost", "empty"], ["empty", "full", "early_discharge"]), ([], [])):
        for e in (eng, ref):
            e.reset(torch.full((n,), 7, dtype=torch.int64))
        obs = [e.set_observation(*attrs) for e in (eng, ref)]
        assert eng.specialized and not ref.specialized
        for e in (eng, ref):
            e.step()
        for _ in range(40):
            for e in (eng, ref):
                e.random_policy(-1, acts, nact)
                e.step(acts, nact)
            torch.cuda.synchronize()
            assert torch.equal(eng.decisions, ref.decisions) and torch.equal(obs[0][0], obs[1][0]) and torch.equal(obs[0][1], obs[1][1])


def test_code_object_of_another_plan_is_rejected():
    from maro_amd import _lib
    from maro_amd.cim import specialize as spec
    from maro_amd.cim.engine import CimBatchEngine
    eng = CimBatchEngine("toy.4p_ssdd_l0.0", 8, durations=50, specialize=False)   # (also when the suite runs with MARO_AMD_SPECIALIZE=1)
    other = spec.plan_defines(eng._cs, _lib.MrxCimConfig(8, 0, 0, 51, 1, 0, 1, 0, 0, 0))
    img = spec.code_object(other)
    buf = ctypes.create_string_buffer(img, len(img))
    assert _lib.load().mrx_cim_load_step_kernels(eng._h, buf, len(img), other.encode()) < 0
    assert b"different plan" in _lib.load().mrx_last_error()
    assert not eng.specialized and np.asarray(eng.step()[2].cpu()).sum() == 0   # still steps with the generic kernels


# ---- citi_bike: mrx_cb_load_step_kernels
def _cb_cases():
    from tests.test_citi_bike_oracle import CASES
    return [c for i, c in enumerate(CASES) if i % 3 == 0]


@pytest.mark.parametrize("case", _cb_cases())
def test_specialized_citi_bike_kernels_reproduce_reference(case, monkeypatch):
    from tests.cb_backend_adapter import CbBackendEnv
    from tests.cb_gpu_backend import CbGpuBackend
    from tests.test_citi_bike_oracle import replay_citi_bike
    monkeypatch.setenv("MARO_AMD_SPECIALIZE", "1")

    def make(data, kw, tt, n_envs=70):
        b = CbGpuBackend(data, n_envs=n_envs, max_actions=1, **kw)
        assert b.eng.specialized
        b.reset(transfer_times=[tt[: b.layout.transfer_times_cap]] * n_envs)   # the reset kernel is specialised too
        return CbBackendEnv(b, env=n_envs - 1)
    replay_citi_bike(make, case)


def test_specialized_citi_bike_equals_generic_batch():
    import torch

    from maro_amd.citi_bike.engine import CitiBikeBatchEngine
    n = 300
    engs = [CitiBikeBatchEngine("toy.5s_6t", n, durations=1500, snapshot_resolution=10, seeds=np.arange(n) + 3, specialize=s) for s in (False, True)]
    assert not engs[0].specialized and engs[1].specialized
    acts = [torch.zeros((n, 1, 3), dtype=torch.int32, device="cuda") for _ in engs]
    nact = [torch.zeros(n, dtype=torch.int32, device="cuda") for _ in engs]
    outs = [e.step() for e in engs]
    for i in range(400):
        for e, a, k in zip(engs, acts, nact):
            e.random_policy(i, a, k)
        outs = [e.step(a, k) for e, a, k in zip(engs, acts, nact)]
        torch.cuda.synchronize()
        for x, y in zip(*outs):
            assert torch.equal(x, y), i
    for name in ("hdr", "live", "ring", "ring_fi"):   # the engines' state views (the raw workspace also holds scratch areas)
        assert torch.equal(getattr(engs[0], name), getattr(engs[1], name)), name


@pytest.mark.parametrize("budget,lanes", [(6, 0), (30, 4)])
def test_specialized_citi_bike_bounded_steps(budget, lanes, monkeypatch):
    """The LDS-frame step kernel under mrx_cb_set_step_budget / mrx_cb_set_lanes_per_wave: trajectories stay the oracle's."""
    from maro_amd.citi_bike.data import load_topology
    from tests.cb_batch_check import run_bounded_vs_oracle
    from tests.cb_gpu_backend import CbGpuBackend
    monkeypatch.setenv("MARO_AMD_SPECIALIZE", "1")
    data = load_topology("toy.5s_6t")
    kw = dict(durations=700, snapshot_resolution=5)
    b = CbGpuBackend(data, n_envs=150, max_actions=1, **kw)
    assert b.eng.specialized
    key_auto = b.eng.code_object_key
    b.eng.set_lanes_per_wave(lanes)
    # a split chosen by hand runs the build that takes the envs-per-wave shift as a kernel argument (the automatic split's is compiled in)
    assert b.eng.specialized and (b.eng.code_object_key != key_auto) == (lanes != 0)
    calls, unready = run_bounded_vs_oracle(b, data, kw, seeds=np.arange(150) + 3, budget=budget, check_envs=[0, 63, 64, 149])
    assert unready > 0


@pytest.mark.parametrize("case", ["cbjoint_tight_m2_alt", "cbjoint_city180_m1_alt"])
def test_specialized_citi_bike_joint_modes(case, monkeypatch):
    """mrx_cb_step_joint on the LDS-frame step kernel."""
    from tests.cb_backend_adapter import CbBackendEnv
    from tests.cb_gpu_backend import CbGpuBackend
    from tests.test_citi_bike_joint_oracle import replay_citi_bike_joint
    monkeypatch.setenv("MARO_AMD_SPECIALIZE", "1")

    def make_joint(data, kw, tt, mode, n_envs=70):
        b = CbGpuBackend(data, n_envs=n_envs, max_actions=1, decision_mode=mode, **kw)
        assert b.eng.specialized
        b.reset(transfer_times=[tt[: b.layout.transfer_times_cap]] * n_envs)
        return CbBackendEnv(b, env=n_envs - 1)
    replay_citi_bike_joint(make_joint, case)


def test_forcing_wave_decisions_on_a_plan_with_a_compiled_in_shift_loads_the_runtime_shift_build():
    """a plan that fits LDS one env per lane gets step kernels with the envs-per-wave shift compiled in — a build without
    the wave replay kernel.  Forcing the wave-cooperative path on (or a lane split by hand) loads the other build first; a fused
    observation then configures, the trajectories stay the same, and switching back restores the first build."""
    import torch

    from maro_amd.citi_bike.engine import CitiBikeBatchEngine
    n = 96
    engs = [CitiBikeBatchEngine("toy.5s_6t", n, durations=1200, snapshot_resolution=10, seeds=np.arange(n) + 3, specialize=True) for _ in range(2)]
    forced = engs[1]
    assert forced.specialized and not forced._forced_wave
    assert forced.set_wave_decisions(1) and forced._forced_wave and forced.specialized
    obs = forced.set_observation(["bikes", "shortage"])
    assert obs is not None and obs.shape[0] == n
    acts = [torch.zeros((n, 1, 3), dtype=torch.int32, device="cuda") for _ in engs]
    nact = [torch.zeros(n, dtype=torch.int32, device="cuda") for _ in engs]
    outs = [e.step() for e in engs]
    for i in range(250):
        for e, a, k in zip(engs, acts, nact):
            e.random_policy(i, a, k)
        outs = [e.step(a, k) for e, a, k in zip(engs, acts, nact)]
        for x, y in zip(*outs):
            assert torch.equal(x, y), i
    forced.set_observation(())
    forced.set_lanes_per_wave(8)          # by hand on top of the forced mode: still the runtime-shift build
    assert forced._manual_lanes and forced.specialized
    forced.set_lanes_per_wave(0)
    assert not forced.set_wave_decisions(0) or True
    assert not forced._forced_wave and not forced._manual_lanes and forced.specialized
    with pytest.raises(Exception):
        forced.set_lanes_per_wave(3)      # not a power of two: refused, and nothing changed
    assert not forced._manual_lanes and forced.specialized
    for i in range(250, 300):
        for e, a, k in zip(engs, acts, nact):
            e.random_policy(i, a, k)
        outs = [e.step(a, k) for e, a, k in zip(engs, acts, nact)]
        for x, y in zip(*outs):
            assert torch.equal(x, y), i
