#!/usr/bin/env python3
"""Known answers for maro_amd/data_lib.py from the REAL reference's readers — ORACLE tooling (needs oracle/build_ref.sh).

Files written by maro_amd.data_lib.write_binary (sorted, duplicated and out-of-order timestamps) are read back with the
reference's own BinaryReader: `items_tick_picker(...)` tick by tick (binary_reader.py:80-112) and `items(start, end, unit)`
(:218-295).  The golden holds the inputs and what the reference yielded; tests/test_data_lib.py replays them through
pick_ticks / items_in_range.  Also proves the writer's files are files the reference accepts.

    python oracle/gen_golden_data_lib.py [--maro /tmp/oracle/maro_src]
"""
import argparse
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--maro", default="/tmp/oracle/maro_src")
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden", "data_lib_reader_kat.json"))
    a = ap.parse_args()
    sys.path.insert(0, REPO)
    import numpy as np

    from maro_amd.data_lib import write_binary
    sys.path.insert(0, a.maro)
    from maro.data_lib.binary_reader import BinaryReader
    rng = np.random.RandomState(20260924)
    cases = []
    for ci in range(12):
        n = int(rng.randint(1, 60))
        unit = str(rng.choice(["s", "m", "h"]))
        us = {"s": 1, "m": 60, "h": 3600}[unit]
        t0 = int(rng.randint(0, 10**6))
        ts = np.sort(t0 + rng.randint(0, 40 * us, n)).astype(np.int64)
        if ci % 3 == 1:                       # a few records out of order (the picker drops / blocks on them)
            for _ in range(max(1, n // 6)):
                i, j = rng.randint(0, n, 2)
                ts[i], ts[j] = ts[j], ts[i]
        start = t0 if ci % 4 else int(ts.min())
        n_ticks = int(rng.randint(5, 50))
        with tempfile.TemporaryDirectory() as tmp:
            p = os.path.join(tmp, "x.bin")
            write_binary(p, {"timestamp": ts, "idx": np.arange(n)}, {"timestamp": "i8", "idx": "i"}, starttime=start, endtime=int(ts.max()))
            reader = BinaryReader(p)
            assert reader.header.item_count == n and reader.start_datetime is not None
            picker = reader.items_tick_picker(0, n_ticks, time_unit=unit)
            got = [-1] * n
            for tick in range(n_ticks):
                for item in picker.items(tick):
                    got[item.idx] = tick
            lo, hi = int(rng.randint(0, 10)), int(rng.randint(10, 45))
            rng_items = [int(item.idx) for item in BinaryReader(p).items(lo, hi, time_unit=unit)]
        cases.append(dict(timestamps=ts.tolist(), starttime=start, n_ticks=n_ticks, unit=unit, ticks=got, range=[lo, hi], range_idx=rng_items))
    with open(a.out, "w") as fp:
        json.dump(dict(source="oracle/gen_golden_data_lib.py: maro.data_lib.binary_reader.BinaryReader on files written by maro_amd.data_lib.write_binary",
                       cases=cases), fp)
    print("golden:", a.out, len(cases), "cases")


if __name__ == "__main__":
    main()
