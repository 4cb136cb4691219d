"""The reference-shaped object API on the real HIP engine."""
import pytest

pytestmark = pytest.mark.gpu


def gpu_factory(topology, n, **kw):
    from maro_amd.cim.engine import CimBatchEngine
    return CimBatchEngine(topology, n, **kw)


def test_vector_env_on_gpu():
    from tests.test_vector_env_api import check_vector_env
    check_vector_env(gpu_factory)


def test_env_view_on_gpu():
    from tests.test_vector_env_api import check_env_view
    check_env_view(gpu_factory)


def test_invalid_action_in_a_batch_on_gpu():
    from tests.test_vector_env_api import check_invalid_action_in_a_batch
    check_invalid_action_in_a_batch(gpu_factory)


def test_joint_decision_modes_object_api_on_gpu():
    from tests.test_vector_env_api import check_joint_object_api
    check_joint_object_api(gpu_factory, "jointseq_toy5p_l05_some")


def test_pipelined_groups_equal_one_engine():
    """PipelinedCimBatch (3 groups on 3 streams) == one engine with the same seeds, env by env."""
    import torch

    from maro_amd.cim.engine import CimBatchEngine
    from maro_amd.cim.rollout import PipelinedCimBatch
    n, topo = 200, "global_trade.22p_l0.8"
    seeds = torch.arange(n, dtype=torch.int64) * 3 + 1
    one = CimBatchEngine(topo, n, durations=60, seeds=seeds)
    bat = PipelinedCimBatch(topo, n, groups=3, seeds=seeds, durations=60)
    a1, n1 = torch.zeros((n, 1, 4), dtype=torch.int32, device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda")
    ag = [torch.zeros((s, 1, 4), dtype=torch.int32, device="cuda") for s in bat.sizes]
    ng = [torch.zeros(s, dtype=torch.int32, device="cuda") for s in bat.sizes]
    one.step()
    bat.for_each(lambda g, e: e.step())
    for _ in range(120):
        one.random_policy(-1, a1, n1)     # keyed on (seed, tick, vessel): independent of the grouping
        one.step(a1, n1)
        bat.for_each(lambda g, e: (e.random_policy(-1, ag[g], ng[g]), e.step(ag[g], ng[g])))
    torch.cuda.synchronize()
    assert torch.equal(bat.cat("decisions"), one.decisions) and torch.equal(bat.cat("metrics"), one.metrics)
    assert torch.equal(bat.cat("done"), one.done) and bool(one.done.all())


def test_object_api_on_a_pipelined_batch_on_gpu():
    """GpuVectorEnv over PipelinedCimBatch (3 HIP engines on 3 streams) == GpuVectorEnv over one engine: VectorEnv.step with
    list / dict actions, snapshot_list slices, tick / frame_index, set_seed + reset, an invalid action in the last group."""
    from maro_amd.cim.rollout import PipelinedCimBatch
    from tests.test_vector_env_api import check_pipelined_object_api

    def grouped(topology, n, **kw):
        return PipelinedCimBatch(topology, n, groups=3, seeds="topology", **kw)
    check_pipelined_object_api(gpu_factory, grouped)


def test_vector_env_groups_argument_on_gpu():
    """GpuVectorEnv(..., groups=3) builds the pipelined batch itself; a whole episode equals the single-engine env."""
    from maro_amd.cim.rollout import PipelinedCimBatch
    from maro_amd.cim.vector_env import GpuVectorEnv
    kw = dict(durations=50, seeds=[3, 4, 5, 6, 7], max_actions=1)
    a, b = GpuVectorEnv(5, "cim", "toy.4p_ssdd_l0.0", **kw), GpuVectorEnv(5, "cim", "toy.4p_ssdd_l0.0", groups=3, **kw)
    assert isinstance(b.engine, PipelinedCimBatch) and b.engine.sizes == [2, 2, 1]
    ra, rb = a.step(None), b.step(None)
    while not ra[2]:
        assert ra[0] == rb[0] and [None if e is None else (e.tick, e.port_idx, e.vessel_idx) for e in ra[1]] == \
            [None if e is None else (e.tick, e.port_idx, e.vessel_idx) for e in rb[1]]
        ra, rb = a.step(None), b.step(None)
    assert rb[2] and ra[0] == rb[0]


def test_whole_batch_step_equals_the_per_env_path_on_gpu():
    from tests.test_vector_env_api import check_whole_batch_step_equals_the_per_env_path
    check_whole_batch_step_equals_the_per_env_path(gpu_factory)
