"""TEST INFRASTRUCTURE — the reference's own citi_bike example agent (``GreedyPolicy``, examples/citi_bike/greedy/launcher.py,
unmodified) driven once by the reference ``Env`` and once by ``GpuVectorEnv(1, "citi_bike", ...).env_view(0)``: every
decision event it sees (tick, station, type, frame index, action scope), every action it takes and the final metrics must be
identical.  Build container only (built reference + the stubs of oracle/gen_golden_citi_bike.py; no GPU here, so the view is
backed by the host-compiled device code, tests/emu).  The reference draws transfer times from the process-global
``np.random`` (decision_strategy.py:213-216): it is seeded with s, the engine env with seeds=[s] (same stream).

    python3 oracle/check_citi_bike_greedy_dropin.py --maro /tmp/oracle/maro_src --stubs /tmp/oracle/stubs
"""
import argparse
import importlib.util
import os
import random
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--maro", default="/tmp/oracle/maro_src")
    ap.add_argument("--stubs", default="/tmp/oracle/stubs")
    ap.add_argument("--topology", default="toy.5s_6t")
    ap.add_argument("--durations", type=int, default=2880)
    ap.add_argument("--resolution", type=int, default=10)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--backend", default="emu", choices=["emu", "gpu"], help="what backs GpuVectorEnv: the host-compiled device code (build container) or libmaro_amd.so on cuda:0")
    args = ap.parse_args()
    os.environ["HOME"] = os.environ.get("MARO_ORACLE_HOME", "/tmp/oracle/home")
    sys.path.insert(0, args.stubs)
    sys.path.insert(0, args.maro)
    sys.path.insert(0, REPO)
    import numpy as np
    from maro.simulator import Env

    from oracle.setup_toy_topologies import ensure_toy
    ensure_toy(args.maro, os.environ["HOME"], args.topology)   # the packaged toy written back (left alone, the reference generates OTHER random toy data)

    spec = importlib.util.spec_from_file_location("greedy_launcher", os.path.join(args.maro, "examples/citi_bike/greedy/launcher.py"))
    greedy = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(greedy)   # class + config only; the script body is under __main__

    from maro_amd.cim.vector_env import GpuVectorEnv
    kw = dict(durations=args.durations, snapshot_resolution=args.resolution)

    def drive(env):
        random.seed(args.seed)
        policy = greedy.GreedyPolicy(2, 2)   # top-2 candidates: random.choice really chooses
        trace = []
        metrics, ev, done = env.step(None)
        while not done:
            action = policy.choose_action(ev)
            trace.append((ev.tick, ev.station_idx, str(ev.type.value), ev.frame_index, tuple(ev.action_scope.items()),
                          (action.from_station_idx, action.to_station_idx, action.number), tuple(sorted(dict(metrics).items()))))
            metrics, ev, done = env.step(action)
        return trace, dict(env.metrics), env.snapshot_list["stations"][::["bikes", "shortage", "fulfillment", "transfer_cost"]]

    np.random.seed(args.seed)
    ref = drive(Env(scenario="citi_bike", topology=args.topology, start_tick=0, **kw))
    if args.backend == "gpu":            # the product path: the HIP engine behind the env view
        ours = drive(GpuVectorEnv(1, "citi_bike", args.topology, max_actions=1, seeds=[args.seed], **kw).env_view(0))
    else:
        from tests.emu.cb_emu_engine import CbEmuEngine
        eng = CbEmuEngine(args.topology, 1, max_actions=1, seeds=[args.seed], **kw)
        ours = drive(GpuVectorEnv(1, "citi_bike", args.topology, _engine=eng, **kw).env_view(0))
    assert len(ref[0]) == len(ours[0]), (len(ref[0]), len(ours[0]))
    for i, (a, b) in enumerate(zip(ref[0], ours[0])):
        assert a == b, (i, a, b)
    assert {k: int(v) for k, v in ref[1].items()} == {k: int(v) for k, v in ours[1].items()}, (ref[1], ours[1])
    assert np.array_equal(np.asarray(ref[2]), np.asarray(ours[2]))
    print(f"OK [{args.backend}]: GreedyPolicy on {args.topology} ({args.durations} ticks, resolution {args.resolution}, seed {args.seed}): "
          f"{len(ref[0])} decision events / actions identical, full stations snapshot history identical, metrics {ref[1]}")


if __name__ == "__main__":
    main()
