#!/usr/bin/env python3
"""Pins the C oracle against the REAL reference on RANDOM topologies — ORACLE tooling (needs oracle/build_ref.sh; log committed
next to this file).  The goldens pin the oracle on 26 hand-picked cases; the device code is fuzzed against the oracle on
hundreds of random topologies (tests/fuzz_topologies.py).  This script closes the triangle: the same random configs
(tests.fuzz_topologies.random_conf: ports, routes that visit a port twice, noisy / zero buffer ticks, both order modes,
random start ticks and snapshot resolutions) are written as config.yml, run through the reference's own Env in a fresh
process each (its RNG registry is process-global), and compared with the oracle decision by decision, metric by metric, and
on the full ports / vessels snapshot tensors at the end.

    python oracle/check_random_topologies.py [first_seed=0] [count=40]
"""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
MARO = os.environ.get("MARO_REFERENCE_BUILD", "/tmp/oracle/maro_src")


def worker(case_seed):
    os.environ.setdefault("HOME", "/tmp/oracle/home")
    os.environ.setdefault("SKIP_DEPLOYMENT", "TRUE")
    sys.path.insert(0, MARO)
    sys.path.insert(0, REPO)   # (the reference checkout has a `tests` package of its own: ours must come first)
    import copy

    import numpy as np
    import yaml
    from maro.simulator import Env
    from maro.simulator.scenarios.cim.common import Action, ActionType

    from maro_amd.cim.topology import parse_config
    from oracle.cim_oracle import CimOracle, hash_policy_action
    from tests.fuzz_topologies import random_conf
    from tests.golden_util import PORT_ATTRS, VESSEL_ATTRS
    def plain(x):   # numpy scalars -> Python (yaml.safe_dump)
        if isinstance(x, dict):
            return {plain(k): plain(v) for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return [plain(v) for v in x]
        return x.item() if isinstance(x, np.generic) else x

    rng = np.random.RandomState(case_seed)
    conf = plain(random_conf(rng))
    conf["stop_number"] = [max(2, int(x)) for x in conf["stop_number"]]   # the reference breaks on 1-slot stop lists (vessel.py:110)
    if case_seed % 2:   # every other config: noise small enough that no noised ratio turns negative — the reference asserts that
        for p in conf["ports"].values():   # every generated order is handed out (cim_data_container.py:396), which negative ratios break
            od = p["order_distribution"]
            od["source"]["noise"] = min(od["source"]["noise"], 0.45 * od["source"]["proportion"])
            for t in od.get("targets", {}).values():
                t["noise"] = min(t["noise"], 0.45 * t["proportion"])
    res, seed = int(rng.choice([1, 1, 2, 3, 5])), int(rng.randint(0, 10**6))
    start = int(rng.choice([0, 0, 1, 2, 3, 13]))   # small start ticks: only SOME vessels' first departure falls before them ("zombie" vessels)
    durations = int(rng.choice([120, 120, 200]))
    ring = None if rng.rand() < 0.6 else int(rng.randint(2, 9))   # a small snapshot ring: eviction order
    folder = tempfile.mkdtemp(prefix="rnd_topo_")
    with open(os.path.join(folder, "config.yml"), "w") as fp:
        yaml.safe_dump(copy.deepcopy(conf), fp, sort_keys=False)
    class RefRaised(Exception):
        pass

    def ref(fn, *a, **k):   # a config the reference itself cannot run (its own asserts / slot-1 bug) is outside the comparison
        try:
            return fn(*a, **k)
        except Exception as e:  # noqa: BLE001
            raise RefRaised(f"{type(e).__name__}: {str(e)[:80]}")

    try:
        env = ref(Env, scenario="cim", topology=folder, start_tick=start, durations=durations, snapshot_resolution=res, max_snapshots=ring)
        env.set_seed(seed)
        ref(env.reset, keep_seed=True)
    except RefRaised as e:
        print(json.dumps(dict(seed=case_seed, skipped="the reference raises on this config: " + str(e))))
        return
    o = CimOracle(parse_config(copy.deepcopy(conf), name="rnd"), start_tick=start, durations=durations, snapshot_resolution=res, max_snapshots=ring)
    o.set_seed(seed)
    o.reset(keep_seed=True)
    mode = [0, 0, 1, 2][case_seed % 4]   # Sequential twice as often as each Joint mode
    if mode:
        from maro.simulator import DecisionMode
        try:
            env = ref(Env, scenario="cim", topology=folder, start_tick=start, durations=durations, snapshot_resolution=res, max_snapshots=ring, decision_mode=DecisionMode(mode))
            env.set_seed(seed)
            ref(env.reset, keep_seed=True)
        except RefRaised as e:
            print(json.dumps(dict(seed=case_seed, skipped="the reference raises on this config: " + str(e))))
            return
    try:
        m, de, done = ref(env.step, None)
    except RefRaised as e:
        print(json.dumps(dict(seed=case_seed, skipped="the reference raises on this config: " + str(e))))
        return
    n = 0
    segments = 2 if (mode == 0 and case_seed % 3 == 0) else 1   # Sequential, every third config: a second episode after reset(keep_seed=False)
    if mode == 0:
      for seg in range(segments):
        if seg:   # cim_data_container_helpers.py:58-60: the new seed is drawn from the route stream as the first episode left it
            try:
                ref(env.reset, keep_seed=False)
                m, de, done = ref(env.step, None)
            except RefRaised as e:
                print(json.dumps(dict(seed=case_seed, steps=n, skipped="the reference raises after reset (compared equal until then): " + str(e))))
                return
            o.reset(keep_seed=False)
        om, od, odone = o.step(None)
        while True:
              assert done == odone, (n, done, odone)
              assert [m["order_requirements"], m["container_shortage"], m["operation_number"]] == [int(x) for x in om], (n, m, om)
              if done:
                  break
              row = [de.tick, de.port_idx, de.vessel_idx, de.action_scope.load, de.action_scope.discharge, de.early_discharge, env.frame_index, 1]
              assert row == [int(x) for x in od], (n, row, od)
              a = hash_policy_action(seed, n, od)
              acts = [a]
              if n % 5 == 0:   # several actions on one decision: the quantity split in two (the same kind of action twice is always legal)
                  acts = [(a[0], a[1], a[2] // 2, a[3]), (a[0], a[1], a[2] - a[2] // 2, a[3])]
              try:
                  m, de, done = ref(env.step, [Action(int(x[0]), int(x[1]), int(x[2]), ActionType.LOAD if x[3] == 0 else ActionType.DISCHARGE) for x in acts])
              except RefRaised as e:
                  print(json.dumps(dict(seed=case_seed, steps=n, skipped="the reference raises mid-episode (compared equal until then): " + str(e))))
                  return
              om, od, odone = o.step(acts)
              n += 1
    else:
        # Joint / JointWithSequentialAction (core.py:354-366), as oracle/gen_golden_joint.py drives it: the first k events answered,
        # DISCHARGE-only wherever a cached scope could be stale (the reference asserts on quantities beyond the live scope)
        import random as pyrandom

        from tests.golden_util import JointPayloadCache
        cache, prng, seen = JointPayloadCache(), pyrandom.Random(case_seed), set()
        om, orows, odone = o.step_joint(mode, None, 0)
        while True:
            assert done == odone, (n, done, odone)
            assert [m["order_requirements"], m["container_shortage"], m["operation_number"]] == [int(x) for x in om], (n, m, om)
            if done:
                break
            des = de
            rows = cache(orows)
            valid = rows[rows[:, 7] == 1]
            assert len(des) == len(valid), (n, len(des), len(valid))
            for i, d in enumerate(des):
                sc = d.action_scope
                assert [d.tick, d.port_idx, d.vessel_idx, sc.load, sc.discharge, d.early_discharge, env.frame_index, 1] == [int(x) for x in valid[i]], (n, i, valid[i])
            k = len(des) if prng.random() < 0.4 else prng.randint(1, len(des))
            ports_here = [d.port_idx for d in des]
            actions, enc = [], []
            for d in des[:k]:
                sc = d.action_scope
                fresh = (d.tick, d.vessel_idx) not in seen and ports_here.count(d.port_idx) == 1
                if prng.random() < 0.5 and sc.load > 0 and fresh:
                    q, ty = prng.randint(0, sc.load), 0
                else:
                    q, ty = prng.randint(0, sc.discharge), 1
                actions.append(Action(d.vessel_idx, d.port_idx, q, ActionType.LOAD if ty == 0 else ActionType.DISCHARGE))
                enc.append((d.vessel_idx, d.port_idx, q, ty))
            seen.update((d.tick, d.vessel_idx) for d in des)
            try:
                m, de, done = ref(env.step, actions)
            except RefRaised as e:
                print(json.dumps(dict(seed=case_seed, steps=n, mode=mode, skipped="the reference raises mid-episode (compared equal until then): " + str(e))))
                return
            om, orows, odone = o.step_joint(mode, enc, k)
            n += len(des)
    sl = env.snapshot_list
    assert sl.get_frame_index_list() == o.frame_indices()
    assert np.array_equal(sl["ports"][::PORT_ATTRS], o.query("ports", [], [], PORT_ATTRS))
    assert np.array_equal(sl["vessels"][::VESSEL_ATTRS], o.query("vessels", [], [], VESSEL_ATTRS))
    print(json.dumps(dict(seed=case_seed, steps=n, mode=mode, start_tick=start, resolution=res, ports=len(conf["ports"]), vessels=len(conf["vessels"]),
                          order_mode=conf.get("order_generate_mode", "fixed"))))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(int(sys.argv[2]))
        return
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    ok, skipped, bad, steps = 0, 0, [], 0
    for s in range(first, first + count):
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", str(s)], capture_output=True, text=True, timeout=600)
        if out.returncode == 0:
            line = out.stdout.strip().splitlines()[-1]
            r = json.loads(line)
            steps += r.get("steps", 0)
            if "skipped" in r:
                skipped += 1
            else:
                ok += 1
            print(line)
        else:
            bad.append(s)
            print("FAILED seed", s, out.stderr.strip().splitlines()[-1][:300])
    print(f"{ok} of {count} random topologies compared in full ({steps} decisions): the C oracle equals the reference (decisions, metrics, ports / vessels "
          f"snapshot history); {skipped} configs the reference itself cannot run; failures: {bad}")


if __name__ == "__main__":
    main()
