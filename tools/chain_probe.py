#!/usr/bin/env python3
"""What bounds the CIM headline loop: the kernels, or the chain of dependent launches?  

A group's rollout step is a chain of DEPENDENT launches on one stream — agent launch (it also carries the schedule block) -> step
kernel — and the next step's agent launch depends on this step's decisions.  Throughput of the whole batch = envs / chain time, as
long as the groups' chains overlap; so a launch gap of a few microseconds per link counts as much as kernel time.  This probe runs the
bench configuration (bench.build_cim_groups: 16384 envs, 3 groups, plan-specialised kernels, fused observation, mid-episode) through
loops that differ only in the number of launches per step:

    agent+step   random-policy launch (+ schedule block) -> sorted step kernel            2 launches   (the headline loop)
    sched+step   no agent (action None); schedule kernel -> sorted step kernel            2 launches
    step1        no agent; unsorted step kernel (launch form 1: no schedule needed)        1 launch
    agent+step1  random-policy launch -> unsorted step kernel                             2 launches

`None` actions leave every decision unanswered (vessels keep sailing, orders keep flowing: the tick work is the same, the action
block of a step is skipped), so "sched+step" vs "agent+step" isolates the agent kernel's own work and "step1" vs "agent+step1" the
cost of one more link in the chain.  Prints one JSON object.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=16384)
    ap.add_argument("--groups", type=int, default=3)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--repeats", type=int, default=3)
    a = ap.parse_args()
    import torch

    import bench
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    out = {"envs": a.envs, "groups": a.groups, "steps": a.steps, "loops": {}}
    for name, agent, mode in (("agent+step", True, 2), ("sched+step", False, 2), ("step1", False, 1), ("agent+step1", True, 1)):
        engines, streams, bufs, sizes, offs = bench.build_cim_groups("global_trade.22p_l0.8", a.envs, a.groups, dev, 0, 1120 + 2 * a.steps * (a.repeats + 2), 4, True, mode,
                                                                     "fused", "random")
        for g, e in enumerate(engines):
            e.reset(torch.arange(sizes[g], dtype=torch.int64) + offs[g] + 1)
            e.step()
        step_i = 1

        def one(i):
            for g, e in enumerate(engines):
                if agent:
                    e.random_policy(i, bufs[g]["actions"], bufs[g]["n_actions"], None)
                    e.step(bufs[g]["actions"], bufs[g]["n_actions"])
                else:
                    e.step()
        # into mid-episode with the real agent (the same state for every variant), then the variant's own warm-up
        for _ in range(700):
            for g, e in enumerate(engines):
                e.random_policy(step_i, bufs[g]["actions"], bufs[g]["n_actions"], None)
                e.step(bufs[g]["actions"], bufs[g]["n_actions"])
            step_i += 1
        for _ in range(50):
            one(step_i)
            step_i += 1
        vals = []
        for _ in range(a.repeats):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(a.steps):
                one(step_i)
                step_i += 1
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            vals.append(dt / a.steps * 1e6)
        valid = sum(int((e.decisions[:, 7] == 1).sum().item()) for e in engines)
        us = sorted(vals)[len(vals) // 2]
        out["loops"][name] = {"us_per_batch_step": us, "env_steps_per_s": valid / (us * 1e-6), "launches_per_group_step": 2 if (agent or mode == 2) else 1,
                              "envs_with_a_decision": valid, "step_mode": engines[0].step_mode}
        del engines, bufs
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
