"""Compile the plan-specialised CIM step kernels (maro_amd/cim/specialize.py) ahead of time into the in-tree cache — no GPU
needed.  `bench`: the plans bench.py uses by default (called by __graft_entry__.build()); `goldens`: every plan the golden
replays create (for a full GPU test run with MARO_AMD_SPECIALIZE=1)."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def plans(which):
    from maro_amd import _lib
    from maro_amd.cim.topology import load_topology

    def cfg(durations, res=1, ring=0, max_actions=1, mode=0, order_table=0, start_tick=0):
        return _lib.MrxCimConfig(1, 0, start_tick, durations, res, ring or 0, max_actions, 0, mode, order_table)

    out = []
    if which in ("bench", "all"):
        t = load_topology("global_trade.22p_l0.8")
        out += [(t, cfg(1120, ring=4)), (t, cfg(1120, ring=8))]            # bench.py default / --policy dqn
    if which in ("tests", "all"):   # tests/test_gpu_step_modes.py, tests/test_gpu_bench_parity.py
        out += [(load_topology("global_trade.22p_l0.8"), cfg(100, ring=5)), (load_topology("toy.5p_ssddd_l0.5"), cfg(100, ring=5)),
                (load_topology("toy.6p_sssbdd_l0.8"), cfg(80, max_actions=6, mode=1)),
                (load_topology("global_trade.22p_l0.8"), cfg(1120, ring=4, max_actions=2))]
    if which in ("goldens", "all"):
        from tests.golden_util import case_topology, golden_cases, joint_golden_cases, load_case, load_joint_case
        for name in golden_cases():
            meta = load_case(name)[1]
            topo, kw = case_topology(meta), meta["kwargs"]
            out.append((topo, cfg(kw["durations"], kw.get("snapshot_resolution", 1), kw.get("max_snapshots"), 2, start_tick=kw.get("start_tick", 0))))
        for name in joint_golden_cases():
            meta = load_joint_case(name)[1]
            topo, kw = case_topology(meta), meta["kwargs"]
            out.append((topo, cfg(kw["durations"], kw.get("snapshot_resolution", 1), kw.get("max_snapshots"), topo.n_vessels, meta["decision_mode"])))
    return out


def main(which="bench"):
    from maro_amd.cim import specialize as spec
    from maro_amd.cim.engine import NODE_ATTRS
    bench_obs = ([NODE_ATTRS["ports"].index(a) for a in ("empty", "full", "on_shipper", "on_consignee", "booking", "shortage", "fulfillment")],
                 [NODE_ATTRS["vessels"].index(a) for a in ("empty", "full", "remaining_space")])   # bench.py QUERY_ATTRS / VESSEL_QUERY_ATTRS
    test_obs = ([NODE_ATTRS["ports"].index(a) for a in ("empty", "full", "shortage", "transfer_cost")],
                [NODE_ATTRS["vessels"].index(a) for a in ("empty", "remaining_space")])
    todo = {}
    for topo, c in plans(which):
        cs = topo.c_struct()
        for order_table in (0, 1) if which != "bench" else (0,):
            c.order_table = order_table
            todo[spec.plan_defines(cs, c)] = 1
            if which in ("bench", "all") and topo.n_ports == 22:
                todo[spec.plan_defines(cs, c, obs=bench_obs)] = 1   # bench.py fuses this observation into the step
            if which in ("tests", "all") and c.decision_mode == 0 and c.max_snapshots == 5:
                todo[spec.plan_defines(cs, c, obs=test_obs)] = 1
    if which in ("bench", "all"):   # bench.py --scenario citi_bike: toy.3s_4t, 4096 envs per GPU
        import numpy as np

        from maro_amd.citi_bike.abi import MrxCbConfig, topology_struct
        from maro_amd.citi_bike.data import load_topology as load_cb
        data = load_cb("toy.3s_4t")
        ts, keep = topology_struct(data)
        cap = data.n_stations * (int((data.time_mean + 6 * data.time_std) / max(data.resolution, 1)) + 2) + 4
        todo[("citi_bike", spec.plan_defines(ts, MrxCbConfig(4096, 0, 0, 44000, 10, 16, 1, cap, 0), "citi_bike"))] = 1
        todo[("citi_bike", spec.plan_defines(ts, MrxCbConfig(32768, 0, 0, 44000, 10, 16, 1, cap, 0), "citi_bike"))] = 1   # config 4 whole on one GPU
        del np, keep
        # ... and city.800s (the reference's own topology size): the bench line of profiles/ (4096 envs, two days) and the plan of
        # tests/test_gpu_citi_bike.py::test_city800_batch_matches_oracle_specialised (300 envs)
        data8 = load_cb("city.800s")
        ts8, keep8 = topology_struct(data8)
        cap8 = data8.n_stations * (int((data8.time_mean + 6 * data8.time_std) / max(data8.resolution, 1)) + 2) + 4
        todo[("citi_bike", spec.plan_defines(ts8, MrxCbConfig(4096, 0, 0, 2880, 10, 16, 1, cap8, 0, 0), "citi_bike"))] = 1
        todo[("citi_bike", spec.plan_defines(ts8, MrxCbConfig(2048, 0, 0, 2880, 10, 16, 1, cap8, 0, 0), "citi_bike"))] = 1   # two env groups per GPU
        todo[("citi_bike", spec.plan_defines(ts8, MrxCbConfig(300, 0, 1440, 130, 10, 6, 1, cap8, 0, 0), "citi_bike"))] = 1
        del keep8
    if which in ("goldens", "all"):   # citi_bike golden replays on the GPU (tests/test_gpu_citi_bike.py, test_gpu_specialized.py: 70 envs)
        import json

        import numpy as np

        from maro_amd.citi_bike.abi import MrxCbConfig, topology_struct
        from maro_amd.citi_bike.data import load_topology as load_cb
        gdir = os.path.join(REPO, "tests", "golden")
        for f in sorted(os.listdir(gdir)):
            if not (f.startswith("cb_") or f.startswith("cbjoint_")):
                continue
            meta = json.loads(bytes(np.load(os.path.join(gdir, f))["meta"]).decode())
            data = load_cb(meta["topology"])
            kw = meta["kwargs"]
            ts, keep = topology_struct(data)
            cap = data.n_stations * (int((data.time_mean + 6 * data.time_std) / max(data.resolution, 1)) + 2) + 4
            cfg = MrxCbConfig(70, 0, 0, kw["durations"], kw.get("snapshot_resolution", 1), kw.get("max_snapshots") or 0, 1, cap, 0, meta.get("decision_mode", 0))
            todo[("citi_bike", spec.plan_defines(ts, cfg, "citi_bike"))] = 1
    with ThreadPoolExecutor(max(1, min(8, os.cpu_count() or 1))) as ex:
        sizes = list(ex.map(lambda d: len(spec.code_object(d[1], scenario=d[0]) if isinstance(d, tuple) else spec.code_object(d)), todo))
    # code objects older than the newest source they were compiled from can never be loaded again (the cache key hashes the
    # sources): drop them, so that a repo snapshot ships the live plans only
    newest = max(os.path.getmtime(os.path.join(spec.CSRC, f)) for unit in spec.UNITS.values() for f in unit[2])
    stale = [f for f in os.listdir(spec.CACHE) if f.endswith(".hsaco") and os.path.getmtime(os.path.join(spec.CACHE, f)) < newest]
    for f in stale:
        os.remove(os.path.join(spec.CACHE, f))
    print(f"spec cache: {len(sizes)} plan(s) ready in {spec.CACHE}" + (f" ({len(stale)} stale code object(s) removed)" if stale else ""), file=sys.stderr)   # stderr: bench.py's stdout is one JSON line


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "bench")
