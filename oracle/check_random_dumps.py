#!/usr/bin/env python3
"""Pins the NATIVE dump-folder reader + the oracle's data mode against the REAL reference on RANDOM topologies — ORACLE tooling
(needs oracle/build_ref.sh; log committed next to this file).  A random config of tests/fuzz_topologies.py is dumped with the
reference's own `dump_from_config` (cim_data_dump.py:259-278), the dump folder is run through the reference's `Env` (data_from_dumps,
cim_data_container_helpers.py:73-123) and through `maro_amd.cim.topology.load_data_folder` + the C oracle, and the two are
compared decision by decision, metric by metric and on the ports / vessels snapshot history; then the engine's device code
(host-compiled on the CPU wave emulator) replays the same compiled dump against the oracle.

    python oracle/check_random_dumps.py [first_seed=0] [count=30]
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
    import numpy as np
    import yaml
    from maro.data_lib.cim import dump_from_config
    from maro.simulator import Env
    from maro.simulator.scenarios.cim.common import Action, ActionType

    from maro_amd.cim.topology import load_data_folder
    from oracle.cim_oracle import CimOracle, hash_policy_action
    from tests.fuzz_topologies import random_conf
    from tests.golden_util import PORT_ATTRS, VESSEL_ATTRS

    def plain(x):
        if isinstance(x, dict):
            return {plain(k): plain(v) for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return [plain(v) for v in x]
        return x.item() if isinstance(x, np.generic) else x

    rng = np.random.RandomState(case_seed)
    conf = plain(random_conf(rng))
    conf["stop_number"] = [max(2, int(x)) for x in conf["stop_number"]]
    for p in conf["ports"].values():   # within the reference's own domain (oracle/check_random_topologies.py)
        od = p["order_distribution"]
        od["source"]["noise"] = min(od["source"]["noise"], 0.45 * od["source"]["proportion"])
        for t in od.get("targets", {}).values():
            t["noise"] = min(t["noise"], 0.45 * t["proportion"])
    durations = int(rng.choice([60, 110]))
    res = int(rng.choice([1, 1, 3]))
    cfg_dir, dump_dir = tempfile.mkdtemp(prefix="rnd_cfg_"), tempfile.mkdtemp(prefix="rnd_dump_")
    with open(os.path.join(cfg_dir, "config.yml"), "w") as fp:
        yaml.safe_dump(conf, fp, sort_keys=False)
    # the dump is made in a process of its own: the reference's RNG registry is process-global, and an Env that loads a dump in the
    # process that generated it gets other per-stream seeds than in a fresh process (sim_random.py:56-71)
    code = ("import sys; sys.path.insert(0, %r); from maro.data_lib.cim import dump_from_config; dump_from_config(%r, %r, %d)"
            % (MARO, os.path.join(cfg_dir, "config.yml"), dump_dir, durations + 40))
    if subprocess.run([sys.executable, "-c", code], capture_output=True, env=dict(os.environ)).returncode != 0:
        print(json.dumps(dict(seed=case_seed, skipped="the reference's dump_from_config raises on this config")))
        return
    try:
        env = Env(scenario="cim", topology=dump_dir, durations=durations, snapshot_resolution=res)
        m, de, done = env.step(None)
    except Exception as e:  # noqa: BLE001
        print(json.dumps(dict(seed=case_seed, skipped=f"the reference raises on this config: {type(e).__name__}")))
        return
    topo = load_data_folder(dump_dir, name="rnd_dump")
    o = CimOracle(topo, durations=durations, snapshot_resolution=res)
    om, od, odone = o.step(None)
    n = 0
    while True:
        assert done == odone, (n, done, odone)
        assert [m["order_requirements"], m["container_shortage"], m["operation_number"]] == [int(x) for x in om], (n, m, om)
        if done:
            break
        row = [de.tick, de.port_idx, de.vessel_idx, de.action_scope.load, de.action_scope.discharge, de.early_discharge, env.frame_index, 1]
        assert row == [int(x) for x in od], (n, row, od)
        a = hash_policy_action(case_seed, n, od)
        try:
            m, de, done = env.step(Action(int(a[0]), int(a[1]), int(a[2]), ActionType.LOAD if a[3] == 0 else ActionType.DISCHARGE))
        except Exception as e:  # noqa: BLE001
            print(json.dumps(dict(seed=case_seed, steps=n, skipped=f"the reference raises mid-episode (compared equal until then): {type(e).__name__}")))
            return
        om, od, odone = o.step([a])
        n += 1
    sl = env.snapshot_list
    assert sl.get_frame_index_list() == o.frame_indices()
    assert np.array_equal(sl["ports"][::PORT_ATTRS], o.query("ports", [], [], PORT_ATTRS))
    assert np.array_equal(sl["vessels"][::VESSEL_ATTRS], o.query("vessels", [], [], VESSEL_ATTRS))
    # ... and the DEVICE code (host-compiled, CPU wave emulator) on the same compiled dump: data_mode 1 of the engine
    from tests.backend_adapter import SingleEnvAdapter
    from tests.emu.emu import EmuBackend
    from tests.golden_util import MATRIX_ATTRS
    o2 = CimOracle(topo, durations=durations, snapshot_resolution=res)
    e = SingleEnvAdapter(EmuBackend(topo, 1, durations=durations, snapshot_resolution=res, max_actions=2))
    om, od, odone = o2.step(None)
    em, ed, edone = e.step(None)
    k = 0
    while True:
        assert odone == edone and np.array_equal(om, em), (k, om, em)
        if odone:
            break
        assert np.array_equal(od, ed), (k, od, ed)
        a = hash_policy_action(case_seed, k, od)
        om, od, odone = o2.step([a])
        em, ed, edone = e.step([a])
        k += 1
    assert k == n and e.frame_indices() == o2.frame_indices()
    for node, attrs in (("ports", PORT_ATTRS), ("vessels", VESSEL_ATTRS), ("matrices", MATRIX_ATTRS)):
        assert np.array_equal(e.query(node, [], [], attrs), o2.query(node, [], [], attrs)), node
    print(json.dumps(dict(seed=case_seed, steps=n, resolution=res, durations=durations, ports=len(conf["ports"]), vessels=len(conf["vessels"]), device_code="equal")))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(int(sys.argv[2]))
        return
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    ok, skipped, bad, steps = 0, 0, [], 0
    for s in range(first, first + count):
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", str(s)], capture_output=True, text=True, timeout=600)
        if out.returncode == 0:
            r = json.loads(out.stdout.strip().splitlines()[-1])
            steps += r.get("steps", 0)
            skipped += "skipped" in r
            ok += "skipped" not in r
            print(json.dumps(r))
        else:
            bad.append(s)
            print("FAILED seed", s, out.stderr.strip().splitlines()[-1][:300])
    print(f"{ok} of {count} random dump folders compared in full ({steps} decisions): native reader + oracle equal the reference, device code (emulator) equals the oracle; {skipped} skipped; failures: {bad}")


if __name__ == "__main__":
    main()
