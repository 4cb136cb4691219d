"""A fixed slice of the citi_bike randomised differential test (tests/fuzz_citi_bike.py): synthetic data sets with up to 40
stations, neighbour filters that cut, host-compiled device code vs the Python oracle."""
import pytest

from tests.fuzz_citi_bike import run_case


@pytest.mark.parametrize("case_seed", [1, 5, 22, 36, 58])
def test_random_citi_bike_data(case_seed):
    assert run_case(case_seed) >= 0
