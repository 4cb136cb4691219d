#!/usr/bin/env python3
"""ORACLE tooling: golden vectors of the REAL ``CIMEnvSampler.eval(num_episodes)`` (maro/rl/rollout/env_sampler.py:561-611 with
examples/cim/rl/env_sampler.py) on the reference ``Env`` with the example's DQN policies in exploit mode: per episode every
interaction's model action and env action (so a replay needs no policy numerics) and the ``info["env_metric"]`` the episode
leaves.  Pinned under tests/golden/sampler_eval_<case>.npz; consumed by tests/test_sampler.py (emulator) and its GPU twin.

    oracle/build_ref.sh && python3 oracle/gen_golden_sampler_eval.py --maro /tmp/oracle/maro_src
"""
import argparse
import json
import os
import sys
from unittest.mock import MagicMock

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = {"toy5p_l05": ("toy.5p_ssddd_l0.5", 200, 3), "gt22p_l08": ("global_trade.22p_l0.8", 150, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--maro", default="/tmp/oracle/maro_src")
    ap.add_argument("--case", default="toy5p_l05")
    args = ap.parse_args()
    topology, durations, episodes = CASES[args.case]
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

        def post_evaluate(self, info_list, ep):   # (the example prints here)
            pass

    sampler = Logged(learn_env=learn_env, test_env=test_env, policies=policies,
                     agent2policy={agent: f"dqn_{agent}.policy" for agent in learn_env.agent_idx_list},
                     reward_eval_delay=reward_shaping_conf["time_window"])
    np.random.seed(11)
    out, bounds = {}, [0]
    res = sampler.eval(num_episodes=episodes)
    # eval() appends deepcopy(self._info) per episode, but the interaction log has no episode marks: split it where the tick restarts
    inter = np.asarray(log, np.int32)
    for i in range(1, len(inter)):
        if inter[i, 6] < inter[i - 1, 6]:
            bounds.append(i)
    bounds.append(len(inter))
    assert len(bounds) == episodes + 1, bounds
    keys = ["order_requirements", "container_shortage", "operation_number"]
    out["env_metric"] = np.asarray([[info["env_metric"][k] for k in keys] for info in res["info"]], np.int64)
    out["interactions"] = inter            # (agent port, model action, vessel, port, quantity, type, tick)
    out["episode_bounds"] = np.asarray(bounds, np.int64)
    out["meta"] = np.frombuffer(json.dumps(dict(case=args.case, topology=topology, durations=durations, episodes=episodes, seed=900, state_dim=state_dim,
                                                metric_keys=keys)).encode(), np.uint8)
    for ep in range(episodes):
        print(f"episode {ep}: {bounds[ep + 1] - bounds[ep]} interactions, env_metric {out['env_metric'][ep].tolist()}")
    path = os.path.join(REPO, "tests", "golden", f"sampler_eval_{args.case}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
