"""A fixed slice of the randomised differential test (tests/fuzz_topologies.py): random topologies within the engine's
limits, device source on the CPU wave emulator vs the C oracle, bit-exact including every snapshot tensor."""
import pytest

from tests.fuzz_topologies import run_case


@pytest.mark.parametrize("case_seed", [3, 7, 19, 42, 77, 101])
def test_random_topology(case_seed):
    run_case(case_seed)
