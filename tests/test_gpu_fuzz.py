"""The randomised differential tests on the real HIP engines (same cases as the CPU harness runs, plus more seeds)."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case_seed", list(range(200, 224)))
def test_random_cim_topology_on_gpu(case_seed):
    from tests.fuzz_topologies import run_case
    from tests.gpu_backend import GpuBackend
    run_case(case_seed, backend=GpuBackend)


@pytest.mark.parametrize("case_seed", list(range(300, 316)))
def test_random_citi_bike_data_on_gpu(case_seed):
    from tests.cb_gpu_backend import CbGpuBackend
    from tests.fuzz_citi_bike import run_case
    run_case(case_seed, backend=CbGpuBackend)
