"""The configuration bench.py times, checked against the oracle over a WHOLE 1120-tick episode (VERDICT r01 "next" item 2):
global_trade.22p_l0.8, 3 groups of 5461/5462 envs on their own streams, plan-specialised kernels, order table, fused
observation, ring of 4 — in the default (sorted) launch form and in the split one."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("step_mode", [0, 4])
def test_bench_configuration_matches_the_oracle_over_a_full_episode(step_mode):
    import torch

    import bench
    from tests.bench_parity import replay_against_oracle
    dev = torch.device("cuda:0")
    engines, streams, bufs, sizes, offs = bench.build_cim_groups("global_trade.22p_l0.8", 16384, 3, dev, 0, 1120, 4, True, step_mode, "fused", "random")
    assert all(e.specialized for e in engines) and engines[0].layout.order_table_on and sizes == [5462, 5461, 5461]
    assert engines[0].step_mode == (step_mode or 2)
    res = replay_against_oracle(engines, bufs, streams, sizes, offs, 0, "global_trade.22p_l0.8", 1120, 66, obs=True)
    assert res["ok"], res
    assert res["envs_checked"] >= 64 and res["env_steps_checked"] > 64 * 2300 and res["observation_checks"] > 64 * 250


def test_reference_golden_of_the_full_episode_on_the_bench_kernels():
    """tests/golden/cim_gt22p_l08_full_rand0.npz (the real reference, 1120 ticks, rand0 agent) replayed on the specialised
    kernels of the bench plan: every decision + metrics + the last frame."""
    from maro_amd.cim.engine import CimBatchEngine
    from tests.backend_adapter import SingleEnvAdapter
    from tests.gpu_backend import GpuBackend
    from tests.test_oracle_golden import replay_case

    def make(topo, kwargs):
        b = GpuBackend.__new__(GpuBackend)
        b.eng = CimBatchEngine(topo, 3, durations=kwargs["durations"], max_snapshots=4, max_actions=2, specialize=True)
        b.topo, b.layout, b.n_envs, b.max_actions, b.max_tick = b.eng.topo, b.eng.layout, 3, 2, kwargs["durations"]
        return SingleEnvAdapter(b, env=1)
    replay_case(make, "gt22p_l08_full_rand0")
