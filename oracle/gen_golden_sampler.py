#!/usr/bin/env python3
"""ORACLE tooling: golden vectors of the REAL maro.rl sampling loop for the CIM example — ``CIMEnvSampler.sample(num_steps)``
(examples/cim/rl/env_sampler.py over maro/rl/rollout/env_sampler.py:438-537) run on the reference ``Env`` with the example's
DQN policies: several calls in a row (the transition cache, the delayed reward — evaluated AFTER the loop for elements at
least `reward_eval_delay` = 99 ticks old, :516-526 — and the per-agent next states carry over between calls).  Recorded:
every interaction's model action and env action (so a replay needs no policy numerics), and every emitted experience
element (tick, agent, state, action, reward, terminal, next_state, next_agent_state).  Pinned under
tests/golden/sampler_<case>.npz; consumed by tests/test_sampler.py (emulator) and its GPU twin.

    oracle/build_ref.sh && python3 oracle/gen_golden_sampler.py --maro /tmp/oracle/maro_src
"""
import argparse
import json
import os
import sys
from unittest.mock import MagicMock

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = {"toy5p_l05": ("toy.5p_ssddd_l0.5", 320, [60, 90, 45, None]), "gt22p_l08": ("global_trade.22p_l0.8", 230, [150, 200, None]),
         # episodes end INSIDE calls with num_steps given: emission at the episode end, cache cleared by _reset, sampling goes on
         "toy5p_l05_rollover": ("toy.5p_ssddd_l0.5", 150, [100, 100, 60, None]),
         # other topology shapes: a hub with six ports (more downstream ports than the state keeps), and short calls that end
         # between two decisions of one tick
         "toy6p_l08": ("toy.6p_sssbdd_l0.8", 260, [7, 33, 120, 1, None]), "toy4p_l00_rollover": ("toy.4p_ssdd_l0.0", 120, [50, 50, 50, None])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--maro", default="/tmp/oracle/maro_src")
    ap.add_argument("--case", default="toy5p_l05")
    args = ap.parse_args()
    topology, durations, calls = CASES[args.case]
    os.environ.setdefault("HOME", "/tmp/oracle/home")
    os.environ.setdefault("SKIP_DEPLOYMENT", "TRUE")
    sys.path.insert(0, args.maro)
    sys.path.insert(0, REPO)
    for name in ["zmq", "zmq.asyncio", "zmq.eventloop", "zmq.eventloop.zmqstream", "tornado", "tornado.ioloop"]:
        sys.modules[name] = MagicMock()
    import torch
    from maro.simulator import Env
    Env(scenario="cim", topology=topology, durations=durations)   # fresh-process stream indices (see check_env_sampler_dropin.py)
    from examples.cim.rl.algorithms.dqn import get_dqn_policy
    from examples.cim.rl.config import action_shaping_conf, reward_shaping_conf, state_dim
    from examples.cim.rl.env_sampler import CIMEnvSampler

    class SeededReset:
        def __init__(self, env, seeds):
            self._env, self._seeds = env, iter(seeds)

        def reset(self, keep_seed=False):
            self._env.set_seed(next(self._seeds))
            self._env.reset(keep_seed=True)

        def __getattr__(self, name):
            return getattr(self._env, name)

    learn_env = SeededReset(Env(scenario="cim", topology=topology, durations=durations), range(100, 200))
    test_env = SeededReset(Env(scenario="cim", topology=topology, durations=durations), range(900, 1000))
    n_ports = len(learn_env.agent_idx_list)
    torch.manual_seed(7)
    policies = [get_dqn_policy(state_dim, len(action_shaping_conf["action_space"]), f"dqn_{i}.policy") for i in range(n_ports)]
    log = []

    class Logged(CIMEnvSampler):
        def _translate_to_env_action(self, action_dict, event):
            out = super()._translate_to_env_action(action_dict, event)
            (port, model_action), = action_dict.items()
            (_, a), = out.items()
            log.append([int(port), int(np.asarray(model_action).reshape(-1)[0]), a.vessel_idx, a.port_idx, a.quantity, 0 if a.action_type.name == "LOAD" else 1, event.tick])
            return out

    sampler = Logged(learn_env=learn_env, test_env=test_env, policies=policies,
                     agent2policy={agent: f"dqn_{agent}.policy" for agent in learn_env.agent_idx_list},
                     reward_eval_delay=reward_shaping_conf["time_window"])
    np.random.seed(11)
    out = {}
    for c, num_steps in enumerate(calls):
        n0 = len(log)
        res = sampler.sample(num_steps=num_steps)
        (exps,) = res["experiences"]
        rows = dict(tick=[], agent=[], state=[], action=[], reward=[], terminal=[], next_state=[], next_agent_state=[])
        for e in exps:
            (agent, st), = e.agent_state_dict.items()
            rows["tick"].append(e.tick); rows["agent"].append(int(agent)); rows["state"].append(np.asarray(st, np.float64))
            rows["action"].append(int(np.asarray(e.action_dict[agent]).reshape(-1)[0])); rows["reward"].append(np.float32(e.reward_dict[agent]))
            rows["terminal"].append(bool(e.terminal_dict[agent])); rows["next_state"].append(np.asarray(e.next_state, np.float64))
            rows["next_agent_state"].append(np.asarray(e.next_agent_state_dict[agent], np.float64))
        for k, v in rows.items():
            out[f"call{c}/{k}"] = np.asarray(v)
        out[f"call{c}/interactions"] = np.array([n0, len(log)], np.int64)
        out[f"call{c}/end_of_episode"] = np.array([int(sampler._end_of_episode)], np.int32)
        out[f"call{c}/env_tick"] = np.array([learn_env.tick], np.int32)
        print(f"call {c}: num_steps={num_steps}: {len(log) - n0} interactions, {len(exps)} experiences emitted, env tick {learn_env.tick}, end_of_episode {sampler._end_of_episode}")
    out["interactions"] = np.asarray(log, np.int32)   # (agent port, model action, vessel, port, quantity, type, tick)
    out["meta"] = np.frombuffer(json.dumps(dict(case=args.case, topology=topology, durations=durations, calls=calls, seed=100,
                                                reward_eval_delay=reward_shaping_conf["time_window"], state_dim=state_dim)).encode(), np.uint8)
    path = os.path.join(REPO, "tests", "golden", f"sampler_{args.case}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
