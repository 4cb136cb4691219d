"""citi_bike object API on the HIP engine (same checks as tests/test_cb_vector_env_api.py)."""
import pytest

from tests.test_cb_vector_env_api import check_vector_env, check_vector_env_joint

pytestmark = pytest.mark.gpu


def gpu_factory(topology, n, **kw):
    from maro_amd.citi_bike.engine import CitiBikeBatchEngine
    return CitiBikeBatchEngine(topology, n, **kw)


def test_vector_env_on_gpu():
    assert check_vector_env(gpu_factory) > 10


@pytest.mark.parametrize("mode", [1, 2])
def test_vector_env_joint_modes_on_gpu(mode):
    assert check_vector_env_joint(gpu_factory, mode) > 10


def test_constructs_its_own_engine():
    from maro_amd import GpuVectorEnv
    env = GpuVectorEnv(5, "citi_bike", "toy.3s_4t", durations=200, snapshot_resolution=10, seeds=[1, 2, 3, 4, 5])
    metrics, events, done = env.step(None)
    assert len(events) == 5 and not done and all(ev.tick == 19 for ev in events)
