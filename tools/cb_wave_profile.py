#!/usr/bin/env python3
"""Where the citi_bike WAVE kernels' time goes (city-size plans): builds the plan-specialised kernels with -DMRX_CB_PROFILE and prints
the mean shader-clock cycles per call and phase of mrx_k_cb_replay_wave and mrx_k_cb_step_wave (cb::WProf).  Tooling only.
    python tools/cb_wave_profile.py [--topology city.800s] [--envs 4096] [--steps 900] [--step-budget 24] [--replay-overlap 0]"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REPLAY = ["state HBM -> LDS", "apply action", "light records (lane 0)", "rebalance sweep", "tick end (pool, next decision, snapshot, frame reset)",
          "action scope + decision row", "tail (no-decision rows, metrics)", "state LDS -> HBM"]
STEP = ["pre (header, masks, next station, scope_wave_ok)", "action + state write-back", "action scope", "-"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--topology", default="city.800s")
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--durations", type=int, default=2880)
    ap.add_argument("--steps", type=int, default=900)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--step-budget", type=int, default=24)
    ap.add_argument("--replay-overlap", type=int, default=0)
    a = ap.parse_args()
    os.environ["MARO_AMD_SPEC_FLAGS"] = (os.environ.get("MARO_AMD_SPEC_FLAGS", "") + " -DMRX_CB_PROFILE").strip()
    import numpy as np
    import torch

    from maro_amd.citi_bike.engine import CitiBikeBatchEngine
    n = a.envs
    eng = CitiBikeBatchEngine(a.topology, n, durations=a.durations, snapshot_resolution=10, max_actions=1, seeds=np.arange(n) + 1, specialize=True)
    assert eng.specialized and eng.set_wave_decisions(0)
    eng.set_step_budget(a.step_budget)
    eng.set_replay_overlap(bool(a.replay_overlap))
    lay = eng.layout
    prof = eng._view(lay.off_prof, (lay.env_stride, 16) if lay.env_major else (16, lay.env_stride))
    prof = prof[:n].T if lay.env_major else prof[:, :n]
    actions = torch.zeros((n, 1, 3), dtype=torch.int32, device=eng.device)
    n_actions = torch.zeros((n,), dtype=torch.int32, device=eng.device)
    eng.step()
    for i in range(1, a.warmup):
        eng.random_policy(i, actions, n_actions)
        eng.step(actions, n_actions)
    torch.cuda.synchronize()
    prof.zero_()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for i in range(a.warmup, a.warmup + a.steps):
        eng.random_policy(i, actions, n_actions)
        eng.step(actions, n_actions)
    ev1.record()
    torch.cuda.synchronize()
    us = ev0.elapsed_time(ev1) / a.steps * 1e3
    c = prof.to(torch.float64).sum(dim=1).cpu().numpy()
    n_replay, n_step = max(c[12], 1), max(c[13], 1)
    print(f"{a.topology}, {n} envs, budget {a.step_budget}, overlap {a.replay_overlap}: {us:.1f} us per batch step (policy + step, profiling build); "
          f"per batch step {c[12] / a.steps:.1f} envs in the replay kernel, {c[13] / a.steps:.1f} in the in-tick kernel")
    print("raw slots per replay call:", " ".join(f"[{k}]={c[k] / n_replay:.1f}" for k in range(16)))
    tot = sum(c[k] for k in (0, 15, 7))
    print(f"mrx_k_cb_replay_wave: {tot / n_replay:.0f} cycles per call")
    for k, name in enumerate(REPLAY):
        print(f"  {c[k] / n_replay:9.0f} cyc  {100 * c[k] / max(tot, 1):5.1f} %  {name}")
    tot = sum(c[k] for k in (8, 9, 10))
    print(f"mrx_k_cb_step_wave: {tot / n_step:.0f} cycles per call (to the end of the action scope)")
    for k, name in enumerate(STEP[:3]):
        print(f"  {c[8 + k] / n_step:9.0f} cyc  {100 * c[8 + k] / max(tot, 1):5.1f} %  {name}")


if __name__ == "__main__":
    main()
