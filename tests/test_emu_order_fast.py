"""The branch-free order-table generator (cim::gen_order_table_fast, CimParams::order_fast) against the generic one
(MRX_ORDER_FAST=0 at planning time keeps it): same device source on the CPU wave emulator, whole tables compared byte for byte,
forward and reverse lane order.  (Against the reference itself the tables are pinned by every golden / oracle replay that runs
with the order table — tests/test_emu_golden.py and friends; this test isolates the generator and sweeps random topologies.)"""
import copy
import ctypes
import os

import numpy as np
import pytest

from maro_amd.cim.topology import load_topology, parse_config
from tests.emu.emu import EmuBackend
from tests.fuzz_topologies import random_conf


def _dims(b):
    buf = ctypes.create_string_buffer(1 << 14)
    assert b._L.emu_dump_dims(b._h, buf, len(buf)) > 0
    return dict((k, int(v)) for k, v in (ln.split() for ln in buf.value.decode().splitlines()))


def _table(topo, n_envs, durations, seeds, fast, reverse=False, start_tick=0):
    old = os.environ.get("MRX_ORDER_FAST")
    os.environ["MRX_ORDER_FAST"] = "1" if fast else "0"
    try:
        b = EmuBackend(topo, n_envs, durations=durations, start_tick=start_tick, reverse=reverse)
    finally:
        if old is None:
            os.environ.pop("MRX_ORDER_FAST")
        else:
            os.environ["MRX_ORDER_FAST"] = old
    d = _dims(b)
    b.reset(np.asarray(seeds, np.int64))
    if not b.layout.order_table_on:
        return d, None
    eb = b.layout.order_elem_bytes
    rows = b.view(b.layout.off_orders, np.uint16 if eb == 2 else np.int32, (n_envs, durations, b.layout.order_row_words)).copy()
    return d, rows


@pytest.mark.parametrize("name,durations", [("global_trade.22p_l0.8", 90), ("global_trade.22p_l0.6", 40), ("toy.4p_ssdd_l0.5", 200),
                                            ("toy.5p_ssddd_l0.8", 200), ("toy.6p_sssbdd_l0.8", 200)])
def test_fast_generator_equals_generic_on_packaged_topologies(name, durations):
    topo = load_topology(name)
    seeds = [4096, 7, 123456789]
    d0, t0 = _table(topo, 3, durations, seeds, fast=False)
    d1, t1 = _table(topo, 3, durations, seeds, fast=True)
    d2, t2 = _table(topo, 3, durations, seeds, fast=True, reverse=True)
    assert d0["order_fast"] == 0 and d1["order_fast"] == 1, (d0["order_fast"], d1["order_fast"])
    assert t0.sum() > 0
    assert np.array_equal(t0, t1) and np.array_equal(t1, t2)


def test_fast_generator_equals_generic_on_random_topologies():
    n_fast = 0
    for case in range(60):
        rng = np.random.RandomState(5000 + case)
        conf = random_conf(rng)
        conf["order_generate_mode"] = "fixed"
        topo = parse_config(copy.deepcopy(conf), name="synthetic")
        start = int(rng.choice([0, 0, 3]))
        seeds = [int(rng.randint(0, 10**6)), 11]
        d1, t1 = _table(topo, 2, 60, seeds, fast=True, reverse=bool(case & 1), start_tick=start)
        if not d1["order_fast"]:
            continue
        n_fast += 1
        d0, t0 = _table(topo, 2, 60, seeds, fast=False, start_tick=start)
        assert d0["order_fast"] == 0
        assert np.array_equal(t0, t1), case
    assert n_fast >= 15, n_fast


def test_plans_that_do_not_qualify_keep_the_generic_generator():
    """No order noise (nothing to draw), a ratio that can turn negative (no uint16 proof, `remaining` may grow), a ratio that can come
    arbitrarily close to zero (no margin for the shared reciprocal): the plan says order_fast = 0 and the generic generator runs."""
    from tests.test_emu_synthetic import base_conf
    d, _ = _table(load_topology("global_trade.22p_l0.0"), 1, 20, [1], fast=True)
    assert d["order_fast"] == 0 and d["order_half"] == 1
    conf = base_conf()
    conf["ports"]["pa"]["order_distribution"]["targets"]["pb"]["noise"] = 0.9   # > its proportion 0.5: the noised ratio can be negative
    d, t = _table(parse_config(copy.deepcopy(conf), name="synthetic"), 1, 40, [5], fast=True)
    assert d["order_fast"] == 0 and d["order_half"] == 0
    conf = base_conf()
    conf["ports"]["pa"]["order_distribution"]["targets"]["pb"]["noise"] = 0.5   # == its proportion: non-negative, but no margin above zero
    d, t = _table(parse_config(copy.deepcopy(conf), name="synthetic"), 1, 40, [5], fast=True)
    assert d["order_fast"] == 0 and d["order_half"] == 1
    conf = base_conf()                                                           # as shipped: qualifies
    d, t = _table(parse_config(copy.deepcopy(conf), name="synthetic"), 1, 40, [5], fast=True)
    assert d["order_fast"] == 1
