#!/usr/bin/env python3
"""Attribute mrx_k_cim_step wave cycles to phases using the -DMRX_PROFILE_PHASES build
(maro_amd/csrc/libmaro_amd_prof.so; built by `hipcc ... -DMRX_PROFILE_PHASES`).  Tooling only."""
import argparse
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
PHASES = ["load", "action", "post_step+snapshot", "land tick inputs (+ pipelined: issue next-env prefetch)", "A order_gen", "B1/B2 depart+returns", "B3 orders",
          "B4 arrivals (load+commit)", "output+predecision snapshot", "store", "B4.a per-vessel reads", "B4.b positions/plans", "header round trip (full path)", "FAST PATH total", "(fast-path step count)"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=16384)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--topology", default="global_trade.22p_l0.8")
    ap.add_argument("--step-mode", type=int, default=0)
    ap.add_argument("--specialized", action="store_true", help="profile the plan-specialised kernel (a code object built with "
                    "-DMRX_PROFILE_PHASES, counters read through mrx_cim_read_kernel_global) instead of libmaro_amd_prof.so")
    args = ap.parse_args()
    import maro_amd._lib as L
    if args.specialized:
        os.environ["MARO_AMD_SPEC_FLAGS"] = (os.environ.get("MARO_AMD_SPEC_FLAGS", "") + " -DMRX_PROFILE_PHASES").strip()
    else:
        L.LIB_PATH = os.path.join(REPO, "maro_amd", "csrc", "libmaro_amd_prof.so")
    import torch
    from maro_amd.cim.engine import CimBatchEngine
    lib = L.load()
    n = args.envs
    eng = CimBatchEngine(args.topology, n, durations=1120, max_snapshots=4, seeds=torch.arange(n) + 1, specialize=args.specialized, step_mode=args.step_mode)
    print("step mode", eng.step_mode)
    if args.specialized:
        eng.set_observation(["empty", "full", "on_shipper", "on_consignee", "booking", "shortage", "fulfillment"], ["empty", "full", "remaining_space"])

        real = lib

        class _Read:   # same call shape as the profile build's mrx_prof_read
            @staticmethod
            def mrx_prof_read(buf, reset):
                L.check(real.mrx_cim_read_kernel_global(eng._h, b"g_mrx_prof", buf, 128, int(reset)), "mrx_cim_read_kernel_global")
        lib = _Read
    actions = torch.zeros((n, 1, 4), dtype=torch.int32, device="cuda")
    nact = torch.zeros((n,), dtype=torch.int32, device="cuda")
    eng.step()
    for i in range(1, args.warmup):
        eng.random_policy(i, actions, nact)
        eng.step(actions, nact)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    lib.mrx_prof_read(buf, 1)
    t0 = eng.ticks.sum().item()
    for i in range(args.warmup, args.warmup + args.steps):
        eng.random_policy(i, actions, nact)
        eng.step(actions, nact)
    torch.cuda.synchronize()
    lib.mrx_prof_read(buf, 0)
    tot = sum(buf[:14])
    waves = n * args.steps
    print(f"{args.topology}: {waves} env-steps, {eng.ticks.sum().item() - t0} ticks; mean {tot / waves:.0f} cycles per env-step")
    nfast = buf[14]
    print(f'  fast-path steps: {nfast} ({100*nfast/waves:.1f} %), {buf[13]/max(nfast,1):.0f} cycles each; full-path steps: {(tot-buf[13])/max(waves-nfast,1):.0f} cycles each')
    print("  raw:", [int(x) // waves for x in buf])
    for name, c in zip(PHASES, buf[:14]):
        print(f"  {name:32s} {c / waves:10.0f} cyc/env-step  {100 * c / tot:5.1f} %")


if __name__ == "__main__":
    main()
