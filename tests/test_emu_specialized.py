"""CIM, plan-specialised build of the device source on the CPU wave emulator (tests/emu/emu.py::build_specialized): every
plan dimension a compile-time constant, exactly what cim_spec.hip compiles for the GPU — the goldens, the online-generator
path and random topologies must replay like the generic build."""
import functools

import pytest

from tests.backend_adapter import SingleEnvAdapter
from tests.emu.emu import EmuBackend
from tests.golden_util import golden_cases
from tests.test_oracle_golden import replay_case

SUBSET = [c for c in golden_cases() if "_full" not in c][::5]


def _make(order_table=0):
    def make(topo, kwargs):
        b = EmuBackend(topo, n_envs=1, start_tick=kwargs.get("start_tick", 0), durations=kwargs["durations"], snapshot_resolution=kwargs.get("snapshot_resolution", 1),
                       max_snapshots=kwargs.get("max_snapshots"), max_actions=2, order_table=order_table, specialized=True)
        return SingleEnvAdapter(b)
    return make


@pytest.mark.parametrize("name", SUBSET)
def test_specialized_emulated_kernels_reproduce_reference(name):
    replay_case(_make(), name)


def test_specialized_online_order_generation_path():
    replay_case(_make(order_table=-1), "toy4p_l00_rand0")


@pytest.mark.parametrize("case_seed", [3, 42, 101])
def test_specialized_random_topology(case_seed):
    from tests.fuzz_topologies import run_case
    run_case(case_seed, backend=functools.partial(EmuBackend, specialized=True))
