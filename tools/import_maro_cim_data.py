#!/usr/bin/env python3
"""Compile a MARO CIM *dump folder* (data_from_dumps) or *real data folder* (data_from_files) into the engine's packaged
topology JSON (maro_amd/cim/topology.py::load_data_folder, data_mode 1 / 2).  Native: the csv layouts and MARO's binary
format are read by maro_amd itself, no MARO checkout needed.

    python tools/import_maro_cim_data.py --folder <dump or real folder> --name my_data --out x.json
    python tools/import_maro_cim_data.py --folder ... --out x.json --check-with-maro /tmp/oracle/maro_src   # (optional, where a built
        reference is importable: the result must equal what the reference's own loaders, cim_data_loader.py:360-450, give)
"""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--folder", required=True)
    ap.add_argument("--name")
    ap.add_argument("--out", required=True)
    ap.add_argument("--check-with-maro", metavar="MARO_ROOT")
    args = ap.parse_args()
    from maro_amd.cim.topology import load_data_folder
    topo = load_data_folder(args.folder, name=args.name)
    with open(args.out, "w") as fp:
        fp.write(topo.to_json())
    print(f"{topo.name}: data_mode {topo.data_mode}, {topo.n_ports} ports, {topo.n_vessels} vessels, {topo.n_targets} order pairs, "
          f"max_tick {topo.data_max_tick}, <= {topo.fixed_max_stops} stops/vessel -> {args.out}")
    if args.check_with_maro:
        os.environ.setdefault("HOME", "/tmp/oracle/home")
        sys.path.insert(0, args.check_with_maro)
        from maro.data_lib.cim.cim_data_loader import load_from_folder, load_real_data_from_folder
        dc = load_real_data_from_folder(args.folder) if topo.data_mode == 2 else load_from_folder(args.folder)
        stops = [[(s.arrival_tick, s.leave_tick) for s in ss] for ss in dc.vessel_stops]
        mine = [[(int(a), int(l)) for a, l in zip(topo.fixed_stops_arrival[v, :n], topo.fixed_stops_leave[v, :n])] for v, n in enumerate(topo.fixed_n_stops)]
        assert stops == mine and list(dc.vessel_period_without_noise) == topo.fixed_vessel_period.tolist() and int(dc.seed) == topo.seed
        print("check-with-maro: stops, vessel periods and seed agree with the reference's loader")


if __name__ == "__main__":
    main()
