#!/usr/bin/env python3
"""Print the LDS plan (MRXC_l_* offsets, bytes per env, workgroups that fit a 160 KiB CU at the measured 1280-byte allocation granularity) of a CIM plan.

    python tools/lds_plan.py [topology] [durations]      (host only: mrx_cim_plan_defines needs no device)
"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import __graft_entry__ as ge
    ge.build()
    from maro_amd import _lib
    from maro_amd.cim import specialize
    from maro_amd.cim.topology import load_topology
    topo = load_topology(sys.argv[1] if len(sys.argv) > 1 else "global_trade.22p_l0.8")
    dur = int(sys.argv[2]) if len(sys.argv) > 2 else 1120
    ts = topo.c_struct()
    cfg = _lib.MrxCimConfig(5461, 0, 0, dur, 1, 4, 1, 0, 0, 0)
    d = dict(re.findall(r"#define MRXC_(\w+) (-?\d+)", specialize.plan_defines(ts, cfg)))
    d = {k: int(v) for k, v in d.items()}
    order = sorted((v, k) for k, v in d.items() if k.startswith("l_") and k not in ("l_mt2", "l_mt3", "l_odelay"))
    for off, k in order:
        print(f"  {k:10s} word {off:6d}  byte {off * 4:6d}")
    print("lean build (registers for the return ring + order quantities):", bool(d.get("lean_ok")))
    for name in ("lds_words", "lds_words_lean", "lds_words_reset", "lds_words_gen"):
        b = d[name] * 4
        g = (b + 1279) // 1280 * 1280   # gfx950 allocates LDS in 1280-byte granules (measured: tools/hbm_pattern_bench --residency)
        print(f"{name}: {b} B -> {g} B allocated -> {163840 // g} workgroups per CU")
    print({k: d[k] for k in ("FW", "PW", "PWH", "ctab_words", "NT", "NTP", "H", "P", "V", "NC", "SMAX", "REC_W", "misc_cap")})


if __name__ == "__main__":
    main()
