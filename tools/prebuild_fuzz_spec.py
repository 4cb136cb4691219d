"""Prebuild (on the CPU, no GPU needed) the plan-specialised code objects that `MARO_AMD_SPECIALIZE=1 python tools/gpu_fuzz_sweep.py
<first> <count>` will ask for, so the GPU box only loads them: the fuzz cases are run with stand-in backends that compute the
plan's defines, compile, and stop.   usage: python tools/prebuild_fuzz_spec.py <first_seed> <count>"""
import os
import sys
from multiprocessing import Pool

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class _Done(Exception):
    pass


def one(seed):
    from maro_amd import _lib
    from maro_amd.cim import specialize as spec

    def cim_backend(topo, n_envs=1, start_tick=0, durations=100, snapshot_resolution=1, max_snapshots=None, max_actions=2,
                    decision_mode=0, order_table=0):
        cfg = _lib.MrxCimConfig(n_envs, 0, start_tick, durations, snapshot_resolution, int(max_snapshots or 0), max_actions, 0,
                                decision_mode, order_table)
        spec.code_object(spec.plan_defines(topo.c_struct(), cfg))
        raise _Done()

    def cb_backend(data, n_envs=1, start_tick=0, durations=100, snapshot_resolution=1, max_snapshots=None, max_actions=1,
                   delivery_capacity=0, transfer_times_cap=0):
        from maro_amd.citi_bike.abi import MrxCbConfig, topology_struct
        if not delivery_capacity:
            delivery_capacity = data.n_stations * (int((data.time_mean + 6 * data.time_std) / max(data.resolution, 1)) + 2) + 4
        ts, keep = topology_struct(data)
        cfg = MrxCbConfig(n_envs, 0, start_tick, durations, snapshot_resolution, int(max_snapshots or 0), max_actions,
                          int(delivery_capacity), int(transfer_times_cap))
        spec.code_object(spec.plan_defines(ts, cfg, "citi_bike"), scenario="citi_bike")
        raise _Done()

    import contextlib
    import io

    from tests.fuzz_citi_bike import run_case as cb_case
    from tests.fuzz_topologies import run_case as cim_case
    n = 0
    for fn, be in ((cim_case, cim_backend), (cb_case, cb_backend)):
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                fn(seed, backend=be)
        except _Done:
            n += 1
        except Exception as e:  # noqa: BLE001  (a case the engine rejects: nothing to prebuild)
            print("seed", seed, "skipped:", repr(e)[:100])
    return n


if __name__ == "__main__":
    first, count = int(sys.argv[1]), int(sys.argv[2])
    with Pool(min(8, os.cpu_count() or 1)) as p:
        done = sum(p.map(one, range(first, first + count)))
    print(f"prebuilt {done} plan(s) for seeds {first}..{first + count - 1}")
