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
