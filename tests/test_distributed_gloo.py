"""N>1 path on CPU: two gloo ranks, each owning a contiguous shard of the global env range (CPU wave emulator as the
engine), trajectories gathered to rank 0 and compared env by env with the oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

TOPO, TOTAL, DUR, STEPS = "toy.5p_ssddd_l0.6", 5, 40, 25


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _hash_policy(seeds):
    from oracle.cim_oracle import hash_policy_action

    def policy(step, dec, done):
        d = dec.numpy()
        acts = np.zeros((len(seeds), 1, 4), np.int32)
        n = np.zeros(len(seeds), np.int32)
        for e in range(len(seeds)):
            if d[e, 7] == 1 and not done[e]:
                acts[e, 0] = hash_policy_action(int(seeds[e]), step, d[e])
                n[e] = 1
        return torch.from_numpy(acts), torch.from_numpy(n)
    return policy


def _worker(rank, world, port, result_path):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from maro_amd.cim.rollout import gather_to_learner, rollout, shard_range
    from tests.emu.emu_engine import EmuEngine
    lo, hi = shard_range(TOTAL, rank, world)
    seeds = np.arange(lo, hi, dtype=np.int64) + 100
    eng = EmuEngine(TOPO, hi - lo, durations=DUR, max_actions=1, seeds=seeds)
    traj = rollout(eng, STEPS, _hash_policy(seeds))
    traj["env_id"] = torch.arange(lo, hi, dtype=torch.int32).expand(STEPS, hi - lo).contiguous()   # env_id i32[n]
    traj["obs"] = (traj["decisions"][:, :, :5].to(torch.float32) * 0.5).contiguous()                  # stands in for obs f32[n, 171]
    sizes = [shard_range(TOTAL, r, world)[1] - shard_range(TOTAL, r, world)[0] for r in range(world)]
    full = gather_to_learner(traj, dst=0, sizes=sizes if world == 4 else None)   # both ways of learning the shard sizes
    if rank == 0:
        torch.save({k: v for k, v in full.items()}, result_path)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_covers_everything():
    from maro_amd.cim.rollout import shard_range
    for total in (1, 5, 16, 17):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


@pytest.mark.parametrize("world", [2, 4])
def test_gloo_rollout_matches_oracle(tmp_path, world):
    """world 2: shards of 3 + 2 envs; world 4: 2 + 1 + 1 + 1 (uneven, one grouped send/recv exchange, no padding)."""
    from oracle.cim_oracle import CimOracle, hash_policy_action
    path = str(tmp_path / "traj.pt")
    mp.spawn(_worker, args=(world, _free_port(), path), nprocs=world, join=True)
    traj = torch.load(path)
    assert traj["env_id"][0].tolist() == list(range(TOTAL)) and traj["obs"].dtype == torch.float32
    assert torch.equal(traj["obs"], traj["decisions"][:, :, :5].to(torch.float32) * 0.5)
    dec, act, done = traj["decisions"].numpy(), traj["actions"].numpy(), traj["done"].numpy()
    assert dec.shape == (STEPS, TOTAL, 8) and act.shape == (STEPS, TOTAL, 1, 4)
    for e in range(TOTAL):
        o = CimOracle(TOPO, durations=DUR)
        o.set_seed(100 + e)
        o.reset(keep_seed=True)
        om, od, odone = o.step(None)
        for t in range(STEPS):
            if odone:
                assert done[t - 1, e] if t else False
                break
            assert np.array_equal(dec[t, e], od), (e, t)
            a = hash_policy_action(100 + e, t, od)
            assert tuple(act[t, e, 0]) == a
            om, od, odone = o.step([a])
            assert np.array_equal(traj["metrics"][t, e].numpy(), om)


# ---- config 5's learner-side collection: every rank runs the batched EnvSampler over its shard, the emitted experiences are
# joined on rank 0 (gather_experiences_to_learner) — and equal what ONE sampler over all envs emits, call by call
S_TOPO, S_TOTAL, S_DUR, S_CALLS = "toy.5p_ssddd_l0.5", 5, 60, (25, 40, 33)


class _Actor:
    def __init__(self, smp):
        self.smp = smp

    def act(self, actions, n_actions, decisions=None, state=None, choice=None):
        from maro_amd.cim.policy import translate_actions
        st = self.smp.state(decisions)
        ma = ((decisions[:, 0] + 3 * decisions[:, 1]) % 21).to(torch.int64)
        translate_actions(ma, decisions, st[:, -1].to(torch.float64), decisions[:, 5], out=actions)
        n_actions[:] = (decisions[:, 7] == 1).to(torch.int32)
        state[:] = st.to(torch.float32)
        choice[:] = ma.to(torch.int32)


def _sample_shard(lo, hi):
    from maro_amd.cim.sampler import CimBatchSampler
    from tests.emu.emu_engine import EmuEngine
    smp = CimBatchSampler(EmuEngine(S_TOPO, hi - lo, durations=S_DUR, max_actions=1, max_snapshots=16), time_window=20)
    seeds = lambda ep: 1000 + 7 * ep + torch.arange(lo, hi, dtype=torch.int64)   # noqa: E731  (a function of the GLOBAL env id)
    return [smp.sample_fused(_Actor(smp), num_steps=k, seeds=seeds, reset_every=4, state_dtype=torch.float64) for k in S_CALLS]


def _sampler_worker(rank, world, port, result_path):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from maro_amd.cim.rollout import gather_experiences_to_learner, shard_range
    lo, hi = shard_range(S_TOTAL, rank, world)
    joined = [gather_experiences_to_learner(res, env_offset=lo, dst=0) for res in _sample_shard(lo, hi)]
    if rank == 0:
        torch.save(joined, result_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_experience_collection_equals_one_sampler(tmp_path, world):
    path = str(tmp_path / "exp.pt")
    mp.spawn(_sampler_worker, args=(world, _free_port(), path), nprocs=world, join=True)
    joined = torch.load(path)
    whole = _sample_shard(0, S_TOTAL)
    total = 0
    for a, b in zip(joined, whole):
        assert set(a) == set(b)
        for key in b:
            assert a[key].dtype == b[key].dtype and torch.equal(a[key], b[key]), key
        total += len(b["tick"])
    assert total > 100


# ---- the return half of config 5's loop: the learner's refreshed networks reach every sampler rank in ONE broadcast of the
# packed blob (rollout.broadcast_policy) and are applied in place; every rank's next act then equals the learner's net
class _PackedCpuActor:
    """CPU stand-in for FusedPerPortDQN: owns a packed weight tensor (fixed address), evaluates the per-port nets from it."""

    def __init__(self, chains):
        from maro_amd.cim.policy import pack_policy
        from tests.test_policy import unpack_index
        self._like, self._pos = chains[0], unpack_index(chains[0])
        self.weights = pack_policy(chains)

    def set_policy_state(self, packed):
        self.weights.copy_(packed)

    def q(self, states, port):
        from maro_amd.cim.policy import PerPortDuelingQNet
        from tests.test_policy import unpack_policy
        chains = [unpack_policy(self.weights[p].numpy(), self._like, self._pos) for p in range(self.weights.shape[0])]
        return PerPortDuelingQNet(chains, 21)(states, port)


def _policy_worker(rank, world, port, result_path):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from maro_amd.cim.policy import pack_policy, random_chains
    from maro_amd.cim.rollout import broadcast_policy
    arch = dict(hidden=(40, 24, 12), head_hidden=20)
    actors = [_PackedCpuActor(random_chains(4, 45, 21, seed=100 + rank, **arch)) for _ in range(2)]   # two env groups per rank, stale nets
    addr = [a.weights.data_ptr() for a in actors]
    new = pack_policy(random_chains(4, 45, 21, seed=7, **arch)) if rank == 0 else None             # only the learner has the update
    got = broadcast_policy(new, actors, src=0)
    assert [a.weights.data_ptr() for a in actors] == addr                                          # in place
    g = torch.Generator().manual_seed(3)
    states, ports = torch.randn(32, 45, generator=g) * 5, torch.randint(0, 4, (32,), generator=g)
    torch.save({"blob": got.clone(), "q": [a.q(states, ports) for a in actors]}, f"{result_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_policy_broadcast_reaches_every_rank(tmp_path):
    from maro_amd.cim.policy import PerPortDuelingQNet, pack_policy, random_chains
    path = str(tmp_path / "pol")
    mp.spawn(_policy_worker, args=(2, _free_port(), path), nprocs=2, join=True)
    learner = random_chains(4, 45, 21, seed=7, hidden=(40, 24, 12), head_hidden=20)
    g = torch.Generator().manual_seed(3)
    states, ports = torch.randn(32, 45, generator=g) * 5, torch.randint(0, 4, (32,), generator=g)
    want_q, want_blob = PerPortDuelingQNet(learner, 21)(states, ports), pack_policy(learner)
    for rank in range(2):
        r = torch.load(f"{path}.{rank}")
        assert torch.equal(r["blob"], want_blob)
        assert all(torch.equal(q, want_q) for q in r["q"])
