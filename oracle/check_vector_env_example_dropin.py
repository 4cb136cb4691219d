"""TEST INFRASTRUCTURE — the reference's VectorEnv example (examples/vector_env/hello.py) run as written, with only the
``VectorEnv`` name bound once to the reference class (one process per env, maro/vector_env/vector_env.py) and once to
``GpuVectorEnv``: the per-step metrics, decision events (all envs), ticks, frame indices and the snapshot slice the example
reads must be identical in both usage modes (push one env forward / push all forward) and across the ``env.reset()`` between
them.  Build container only; the engine side is backed by the CPU wave emulator (no GPU here).

    python3 oracle/check_vector_env_example_dropin.py --maro /tmp/oracle/maro_src
"""
import argparse
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--maro", default="/tmp/oracle/maro_src")
    ap.add_argument("--backend", default="emu", choices=["emu", "gpu"], help="what backs GpuVectorEnv: the CPU wave emulator (build container) or libmaro_amd.so on cuda:0")
    ap.add_argument("--stubs", default=None)
    args = ap.parse_args()
    os.environ.setdefault("HOME", "/tmp/oracle/home")
    sys.path.insert(0, args.maro)
    if args.stubs:
        sys.path.insert(1, args.stubs)
    sys.path.insert(0, REPO)
    import numpy as np
    from maro.vector_env import VectorEnv as RefVectorEnv

    from maro_amd.cim.vector_env import GpuVectorEnv

    src = open(os.path.join(args.maro, "examples/vector_env/hello.py")).read()
    body = src[src.index('if __name__ == "__main__":'):].replace('if __name__ == "__main__":', "if True:")
    # the only edit: record what the example's own loop sees after every env.step (inserted after that line)
    body, n = re.subn(r"(\n\s+)(metrics, decision_event, is_done = env\.step\(action\))", r"\1\2\1TRACE(env, metrics, decision_event, is_done, locals().get('ss0'))", body)
    assert n == 1
    head = src[:src.index('if __name__ == "__main__":')]

    def run(vector_env_cls):
        trace = []

        def TRACE(env, metrics, events, done, ss0):
            evs = [None if e is None else (e.tick, e.port_idx, e.vessel_idx, e.action_scope.load, e.action_scope.discharge, e.early_discharge) for e in events] if events else events
            mets = [None if m is None else {k: int(v) for k, v in dict(m).items()} for m in metrics] if metrics else metrics
            trace.append((mets, evs, bool(done), list(env.tick), list(env.frame_index), None if ss0 is None else np.asarray(ss0).tolist()))

        ns = {"__name__": "example", "TRACE": TRACE}
        exec(compile(head, "hello_head", "exec"), ns)
        ns["VectorEnv"] = vector_env_cls
        exec(compile(body, "hello_body", "exec"), ns)
        return trace

    def ours(batch_num, scenario, topology, durations):
        if args.backend == "gpu":        # the product path: the HIP engine
            return GpuVectorEnv(batch_num, scenario, topology, durations=durations, max_actions=1)
        from tests.emu.emu_engine import EmuEngine
        eng = EmuEngine(topology, batch_num, durations=durations, max_actions=1)
        return GpuVectorEnv(batch_num, scenario, topology, durations=durations, _engine=eng)

    ref = run(RefVectorEnv)
    got = run(ours)
    assert len(ref) == len(got), (len(ref), len(got))
    for i, (a, b) in enumerate(zip(ref, got)):
        assert a == b, (i, a, b)
    print(f"OK [{args.backend}]: examples/vector_env/hello.py, both usage modes: {len(ref)} VectorEnv.step calls with identical metrics, decision "
          f"events of all 4 envs, ticks, frame indices and snapshot slices")


if __name__ == "__main__":
    main()
