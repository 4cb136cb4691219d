#!/usr/bin/env python3
"""The synthetic REAL-DATA folder of tests/test_data_lib.py (csv, and MARO binary written by maro_amd.data_lib.write_binary) through
the REAL reference's `Env` (data_from_files, cim_data_container_helpers.py:126-150) against the native `load_data_folder` + the
oracle's data mode 2 — ORACLE tooling (needs oracle/build_ref.sh).  Proves the writer's stops.bin / orders.bin are files the
reference's loader accepts, and that both readers see the same data.   python oracle/check_real_data_folder.py"""
import subprocess
import sys

CODE = r'''
import os, sys, tempfile
os.environ.setdefault("HOME", "/tmp/oracle/home"); os.environ.setdefault("SKIP_DEPLOYMENT", "TRUE")
sys.path.insert(0, os.environ.get("MARO_REFERENCE_BUILD", "/tmp/oracle/maro_src")); sys.path.insert(0, sys.argv[2])
import numpy as np
from maro.simulator import Env
from maro.simulator.scenarios.cim.common import Action, ActionType
from maro_amd.cim.topology import load_data_folder
from oracle.cim_oracle import CimOracle, hash_policy_action
from tests.test_data_lib import _write_cim_folder
from tests.golden_util import PORT_ATTRS, VESSEL_ATTRS
folder = os.path.join(tempfile.mkdtemp(), "f")
_write_cim_folder(folder, sys.argv[1] == "bin")
env = Env(scenario="cim", topology=folder, durations=40)
o = CimOracle(load_data_folder(folder, name="x"), durations=40)
m, de, done = env.step(None); om, od, odone = o.step(None); n = 0
while True:
    assert done == odone
    assert [m["order_requirements"], m["container_shortage"], m["operation_number"]] == [int(x) for x in om], (n, m, om)
    if done: break
    row = [de.tick, de.port_idx, de.vessel_idx, de.action_scope.load, de.action_scope.discharge, de.early_discharge, env.frame_index, 1]
    assert row == [int(x) for x in od], (n, row, od)
    a = hash_policy_action(3, n, od)
    m, de, done = env.step(Action(int(a[0]), int(a[1]), int(a[2]), ActionType.LOAD if a[3] == 0 else ActionType.DISCHARGE))
    om, od, odone = o.step([a]); n += 1
sl = env.snapshot_list
assert np.array_equal(sl["ports"][::PORT_ATTRS], o.query("ports", [], [], PORT_ATTRS))
assert np.array_equal(sl["vessels"][::VESSEL_ATTRS], o.query("vessels", [], [], VESSEL_ATTRS))
print("real-data folder (%s): reference Env == native loader + oracle over %d decisions and the full snapshot history" % (sys.argv[1], n))
'''

if __name__ == "__main__":
    import os
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for kind in ("csv", "bin"):
        out = subprocess.run([sys.executable, "-c", CODE, kind, repo], capture_output=True, text=True)
        print(out.stdout.strip().splitlines()[-1] if out.returncode == 0 else "FAILED " + out.stderr.strip().splitlines()[-1][:300])
