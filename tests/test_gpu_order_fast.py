"""cim::gen_order_table_fast on the GPU: the per-port shared reciprocal (`of_recip` / `of_div`) only exists in the device build —
the host build of the same source divides — so the fast generator's tables are compared here, byte for byte, with the generic
generator's (MRX_ORDER_FAST=0 at planning time) on packaged and random topologies, generic and plan-specialised kernels.  Both are
pinned to the reference by the golden replays."""
import copy
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _table(topo, n_envs, durations, seeds, fast, specialize=False, start_tick=0):
    from maro_amd.cim.engine import CimBatchEngine
    old = os.environ.get("MRX_ORDER_FAST")
    os.environ["MRX_ORDER_FAST"] = "1" if fast else "0"
    try:
        eng = CimBatchEngine(topo, n_envs, durations=durations, start_tick=start_tick, specialize=specialize, seeds=torch.as_tensor(seeds, dtype=torch.int64))
    finally:
        if old is None:
            os.environ.pop("MRX_ORDER_FAST")
        else:
            os.environ["MRX_ORDER_FAST"] = old
    lay = eng.layout
    if not lay.order_table_on:
        return None
    torch.cuda.synchronize()
    dt = torch.int16 if lay.order_elem_bytes == 2 else torch.int32
    return eng._view(lay.off_orders, dt, (n_envs, durations, lay.order_row_words)).cpu().numpy().copy()


def _is_fast(topo, durations, start_tick=0):
    import re

    from maro_amd import _lib
    from maro_amd.cim import specialize as spec
    d = spec.plan_defines(topo.c_struct(), _lib.MrxCimConfig(1, 0, start_tick, durations, 1, 0, 1, 0, 0, 0))
    return int(dict(re.findall(r"#define MRXC_(\w+) (\S+)", d))["order_fast"])


@pytest.mark.parametrize("name,durations,specialize", [("global_trade.22p_l0.8", 1120, True), ("global_trade.22p_l0.8", 300, False),
                                                        ("global_trade.22p_l0.5", 200, False), ("toy.5p_ssddd_l0.8", 400, True),
                                                        ("toy.6p_sssbdd_l0.6", 400, False)])
def test_fast_generator_equals_generic(name, durations, specialize):
    from maro_amd.cim.topology import load_topology
    topo = load_topology(name)
    assert _is_fast(topo, durations) == 1
    seeds = np.arange(48, dtype=np.int64) * 7919 + 13
    t0 = _table(topo, 48, durations, seeds, fast=False, specialize=specialize)
    t1 = _table(topo, 48, durations, seeds, fast=True, specialize=specialize)
    assert t0.sum() != 0
    assert np.array_equal(t0, t1)


def test_fast_generator_equals_generic_on_random_topologies():
    from maro_amd.cim.topology import parse_config
    from tests.fuzz_topologies import random_conf
    n_fast = 0
    for case in range(120):
        rng = np.random.RandomState(9000 + case)
        conf = random_conf(rng)
        conf["order_generate_mode"] = "fixed"
        topo = parse_config(copy.deepcopy(conf), name="synthetic")
        start = int(rng.choice([0, 0, 3]))
        if not _is_fast(topo, 150, start):
            continue
        n_fast += 1
        seeds = rng.randint(0, 10**6, 16).astype(np.int64)
        t0 = _table(topo, 16, 150, seeds, fast=False, start_tick=start)
        t1 = _table(topo, 16, 150, seeds, fast=True, start_tick=start)
        assert np.array_equal(t0, t1), case
    assert n_fast >= 20, n_fast
