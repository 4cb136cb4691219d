"""Batched observation / reward shaping (maro_amd/cim/sampler.py) against vectors produced by the reference's own
snapshot_list with the shaping code of examples/cim/rl/env_sampler.py (oracle/gen_golden.py, case *_sampler).
State vectors are exact (float64 copies of integer state); the reward is a float32 of a 99-term float64 dot
product, compared with rtol 1e-6 (summation order of torch.matmul vs numpy.dot)."""
import numpy as np
import pytest
import torch

from maro_amd.cim.sampler import CimBatchSampler
from tests.golden_util import case_topology, load_case, segment_actions


def run_sampler_case(engine_factory, n_envs=3):
    z, meta = load_case("toy5p_l05_sampler")
    topo = case_topology(meta)
    dur = meta["kwargs"]["durations"]
    eng = engine_factory(topo, n_envs, durations=dur, max_actions=2, seeds=[topo.seed] * n_envs)
    smp = CimBatchSampler(eng)
    gd, gs, gr = z["seg0/decisions"], z["seg0/sampler_state"], z["seg0/sampler_reward"]
    acts = segment_actions(z, 0)
    assert smp.state_dim == gs.shape[1] == 171
    dec, met, done = eng.step()
    A = 2
    for i in range(len(gd)):
        d = dec.cpu().numpy()
        assert np.array_equal(d[0], gd[i]) and np.array_equal(d[-1], gd[i])
        st = smp.state().cpu().numpy()
        assert st.dtype == np.float64 and np.array_equal(st[0], gs[i]) and np.array_equal(st[n_envs - 1], gs[i]), i
        a = np.full((n_envs, A, 4), -1, np.int32)
        na = np.zeros(n_envs, np.int32)
        for j, x in enumerate(acts[i]):
            a[:, j] = x
        na[:] = len(acts[i])
        dec, met, done = eng.step(a, na)
    assert bool(done.cpu().numpy().all())
    # delayed rewards of all decisions, evaluated at the end of the episode like AbsEnvSampler does
    dev = eng.decisions.device
    for i0 in range(0, len(gd), n_envs):
        rows = list(range(i0, min(i0 + n_envs, len(gd))))
        rows = rows + [rows[-1]] * (n_envs - len(rows))
        tick = torch.tensor(gd[rows, 0], dtype=torch.int32, device=dev)
        port = torch.tensor(gd[rows, 1], dtype=torch.int32, device=dev)
        r = smp.reward(tick, port).cpu().numpy()
        assert r.dtype == np.float32
        np.testing.assert_allclose(r, gr[rows], rtol=1e-6, atol=1e-6)


def emu_factory(topology, n, **kw):
    from tests.emu.emu_engine import EmuEngine
    return EmuEngine(topology, n, **kw)


def test_sampler_on_emulator():
    run_sampler_case(emu_factory)


@pytest.mark.gpu
def test_sampler_on_gpu():
    from maro_amd.cim.engine import CimBatchEngine
    run_sampler_case(lambda topo, n, **kw: CimBatchEngine(topo, n, **kw), n_envs=5)
