#!/usr/bin/env python3
"""Where a citi_bike lane's step goes: builds the plan-specialised step kernel with -DMRX_CB_PROFILE (shader-clock cycles per
phase, summed per env in the workspace's prof array) and prints the mean cycles per env-step and phase.  Tooling only.
    python tools/cb_phase_profile.py [--topology toy.3s_4t] [--envs 4096] [--steps 300]"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
PHASES = ["header + frame load, stream window open", "apply action", "light records (returns, trips) incl. waiting for the wave's longest run",
          "rebalance check (station sweep)", "end of tick (late deliveries, snapshot, frame reset)", "waiting for the wave at the loop exit",
          "decision output + end of action_scope (scope rows out)", "-", "-", "scope: neighbour list", "scope: requirements select", "scope: trip-window frame rows",
          "scope: trip-window sums", "scope: trip-window select"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--topology", default="toy.3s_4t")
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=50)
    a = ap.parse_args()
    os.environ["MARO_AMD_SPEC_FLAGS"] = (os.environ.get("MARO_AMD_SPEC_FLAGS", "") + " -DMRX_CB_PROFILE").strip()
    import numpy as np
    import torch

    from maro_amd.citi_bike.data import load_topology
    from maro_amd.citi_bike.engine import CitiBikeBatchEngine
    n = a.envs
    data = load_topology(a.topology)
    eng = CitiBikeBatchEngine(data, n, durations=min(44000, len(data.tick_day)), snapshot_resolution=10, max_snapshots=16, max_actions=1,
                              seeds=np.arange(n) + 1, specialize=True)
    prof = eng._view(eng.layout.off_prof, (16, eng.layout.env_stride))[:, :n]
    actions = torch.zeros((n, 1, 3), dtype=torch.int32, device=eng.device)
    n_actions = torch.zeros((n,), dtype=torch.int32, device=eng.device)
    eng.step()
    for i in range(1, a.warmup):
        eng.random_policy(i, actions, n_actions)
        eng.step(actions, n_actions)
    torch.cuda.synchronize()
    prof.zero_()
    t0 = eng.ticks.to(torch.int64).sum().item()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for i in range(a.warmup, a.warmup + a.steps):
        eng.random_policy(i, actions, n_actions)
        eng.step(actions, n_actions)
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / a.steps
    c = prof.to(torch.float64).sum(dim=1).cpu().numpy() / (n * a.steps)
    tot = c.sum()
    print(f"{a.topology}, {n} envs: {ms * 1e3:.1f} us per batch step (policy + step), {(eng.ticks.to(torch.int64).sum().item() - t0) / (n * a.steps):.2f} ticks per env-step; "
          f"{tot:.0f} cycles per env-step attributed")
    for k, name in enumerate(PHASES):
        print(f"  {c[k]:9.0f} cyc  {100 * c[k] / max(tot, 1):5.1f} %  {name}")


if __name__ == "__main__":
    main()
