"""TEST INFRASTRUCTURE — the drop-in claim checked with the REAL maro.rl code: the reference's own ``CIMEnvSampler``
(examples/cim/rl/env_sampler.py over maro/rl/rollout/env_sampler.py) is run twice with identical random-init DQN
policies and RNG seeds, once on the reference ``Env`` and once on ``GpuVectorEnv(...).env_view(0)``; every experience
element it collects (states, actions, rewards, terminal flags, next states) and the env metrics must be identical.

Runs in the build container only (needs the built reference, see oracle/gen_golden.py; no GPU here, so the env view is
backed by the CPU wave emulator running the device source — tests/emu).  zmq/tornado (distributed training proxies of
maro.rl, absent offline) are stubbed; nothing of them is on this path.

    python3 oracle/check_env_sampler_dropin.py --maro /tmp/oracle/maro_src [--topology toy.5p_ssddd_l0.5 --durations 150]
"""
import argparse
import os
import sys
from unittest.mock import MagicMock

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--maro", default="/tmp/oracle/maro_src")
    ap.add_argument("--topology", default="toy.5p_ssddd_l0.5")
    ap.add_argument("--durations", type=int, default=150)
    ap.add_argument("--episodes", type=int, default=2)
    ap.add_argument("--backend", default="emu", choices=["emu", "gpu"], help="what backs GpuVectorEnv: the CPU wave emulator (build container) or libmaro_amd.so on cuda:0")
    ap.add_argument("--stubs", default=None, help="folder of import stand-ins for packages the reference wants and the box lacks (holidays, geopy)")
    args = ap.parse_args()
    os.environ.setdefault("HOME", "/tmp/oracle/home")
    sys.path.insert(0, args.maro)
    if args.stubs:
        sys.path.insert(1, args.stubs)
    sys.path.insert(0, REPO)
    for name in ["zmq", "zmq.asyncio", "zmq.eventloop", "zmq.eventloop.zmqstream", "tornado", "tornado.ioloop"]:
        sys.modules[name] = MagicMock()
    import torch
    from maro.simulator import Env

    # The reference numbers its named random streams in order of first use, process-wide (sim_random.py:56-71), and
    # importing examples.cim.rl builds toy.4p_ssdd_l0.0 Envs (its rl_component_bundle) that touch route_init before
    # order_init.  A throwaway Env of OUR topology first gives the streams their fresh-process indices — the semantics the
    # engine (and the reference's one-process-per-env VectorEnv) has.
    Env(scenario="cim", topology=args.topology, durations=args.durations)
    from examples.cim.rl.algorithms.dqn import get_dqn_policy
    from examples.cim.rl.config import action_shaping_conf, reward_shaping_conf
    from examples.cim.rl.env_sampler import CIMEnvSampler

    from maro_amd.cim.vector_env import GpuVectorEnv

    def make_ref():
        return Env(scenario="cim", topology=args.topology, durations=args.durations)

    def make_ours():
        if args.backend == "gpu":        # the product path: the HIP engine behind the env view
            return GpuVectorEnv(1, "cim", args.topology, durations=args.durations, max_actions=1).env_view(0)
        from tests.emu.emu_engine import EmuEngine
        eng = EmuEngine(args.topology, 1, durations=args.durations, max_actions=1)
        return GpuVectorEnv(1, "cim", args.topology, durations=args.durations, _engine=eng).env_view(0)

    class SeededReset:
        """The reference keeps its random streams in a process-global registry (maro/simulator/utils/sim_random.py), so two
        Envs in one process — the sampler's learn_env and test_env — perturb each other's seed re-draw at reset; the engine
        has fresh-process semantics per env (as the reference's own VectorEnv, one process per env).  Both sides therefore
        reset through set_seed(s_k) + reset(keep_seed=True) with the same explicit seed sequence."""

        def __init__(self, env, seeds):
            self._env, self._seeds = env, iter(seeds)

        def reset(self, keep_seed=False):
            seed = next(self._seeds)
            if os.environ.get("DROPIN_DEBUG"):
                print("reset ->", seed, type(self._env).__name__)
            self._env.set_seed(seed)
            self._env.reset(keep_seed=True)

        def __getattr__(self, name):
            return getattr(self._env, name)

    def run(make_env, first_seed=100):
        learn_env, test_env = SeededReset(make_env(), range(first_seed, first_seed + 100)), SeededReset(make_env(), range(900, 1000))
        n_ports = len(learn_env.agent_idx_list)
        from examples.cim.rl.config import state_dim   # (look_back + 1) * (max_ports_downstream + 1) * 7 + 3 = 171
        torch.manual_seed(7)
        policies = [get_dqn_policy(state_dim, len(action_shaping_conf["action_space"]), f"dqn_{i}.policy") for i in range(n_ports)]
        sampler = CIMEnvSampler(learn_env=learn_env, test_env=test_env, policies=policies,
                                agent2policy={agent: f"dqn_{agent}.policy" for agent in learn_env.agent_idx_list},
                                reward_eval_delay=reward_shaping_conf["time_window"])
        np.random.seed(11)
        out = []
        for _ in range(args.episodes):
            res = sampler.sample()
            out.append((res["experiences"], dict(sampler._info["env_metric"])))
        np.random.seed(12)
        ev = sampler.eval()
        return out, ev

    def flat(x):
        if isinstance(x, dict) or (hasattr(x, "items") and hasattr(x, "keys")):   # dict or the reference's DocableDict
            return [(k, flat(v)) for k, v in sorted(x.items(), key=lambda kv: str(kv[0]))]
        if isinstance(x, (list, tuple)):
            return [flat(v) for v in x]
        if isinstance(x, np.ndarray):
            return ("nd", x.dtype.str, x.shape, x.tobytes())
        if isinstance(x, np.generic):
            return x.item()
        if hasattr(x, "__dict__") and not callable(x):
            return (type(x).__name__, flat({k: v for k, v in vars(x).items() if k not in ("event",)}))
        return x

    ref, ref_eval = run(make_ref)
    ours, ours_eval = run(make_ours, 101 if os.environ.get("DROPIN_PERTURB") else 100)   # DROPIN_PERTURB: the check must fail
    n_exp = n_rows = 0
    for ep, ((re, rm), (oe, om)) in enumerate(zip(ref, ours)):
        assert rm == om, (ep, rm, om)
        assert len(re) == len(oe), (ep, len(re), len(oe))
        for i, (a, b) in enumerate(zip(re, oe)):
            assert len(a) == len(b)
            for j, (x, y) in enumerate(zip(a, b)):
                assert flat(x) == flat(y), f"episode {ep} batch {i} element {j} differs"
                n_exp += 1
                n_rows += sum(len(np.atleast_1d(v)) for v in getattr(x, "action_dict", {}).values())
    assert flat(ref_eval["info"]) == flat(ours_eval["info"]), (ref_eval["info"], ours_eval["info"])
    print(f"OK [{args.backend}]: {args.episodes} sampled episode(s) + 1 evaluation episode of {args.topology} ({args.durations} ticks): "
          f"{n_exp} experience elements ({n_rows} action rows) identical (states, actions, rewards, terminals, next states), "
          f"env metrics identical: {ref[-1][1]}; eval metrics {ref_eval['info']}")


if __name__ == "__main__":
    main()
