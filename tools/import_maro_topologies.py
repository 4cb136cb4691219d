#!/usr/bin/env python3
"""Compile the CIM topologies of a MARO checkout into the engine's packaged flat-array JSON form.

    python tools/import_maro_topologies.py /path/to/maro [--out maro_amd/cim/topologies]

Reads ``<maro>/maro/simulator/scenarios/cim/topologies/*/config.yml`` (data, not code), runs them
through ``maro_amd.cim.topology.parse_config`` and writes one ``<name>.json`` per topology.  The
packaged files let ``Env``-style construction by topology *name* work where no MARO checkout exists
(e.g. on the GPU box); a filesystem path to a ``config.yml`` folder is always accepted as well.
"""
import argparse
import glob
import os
import sys

import yaml

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from maro_amd.cim.topology import parse_config  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("maro_root")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "maro_amd", "cim", "topologies"))
    args = ap.parse_args()
    pattern = os.path.join(args.maro_root, "maro", "simulator", "scenarios", "cim", "topologies", "*", "config.yml")
    os.makedirs(args.out, exist_ok=True)
    n = 0
    for path in sorted(glob.glob(pattern)):
        name = os.path.basename(os.path.dirname(path))
        with open(path) as fp:
            topo = parse_config(yaml.safe_load(fp), name=name)
        with open(os.path.join(args.out, name + ".json"), "w") as fp:
            fp.write(topo.to_json())
        n += 1
    print(f"compiled {n} topologies into {os.path.abspath(args.out)}")


if __name__ == "__main__":
    main()
