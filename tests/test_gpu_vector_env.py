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


def test_joint_decision_modes_object_api_on_gpu():
    from tests.test_vector_env_api import check_joint_object_api
    check_joint_object_api(gpu_factory, "jointseq_toy5p_l05_some")
