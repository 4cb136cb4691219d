"""Sharded rollouts: independent env batches per GPU, trajectories gathered to the learner rank.

Envs never interact (separate ``Env`` objects / processes in the reference: vector_env/env_process.py:32), so the
env index range is split contiguously across ranks — one process per GPU, its own engine and stream — and the
data path needs NO collective.  The only exchange is the per-rollout gather of trajectory tensors to the
learner (``torch.distributed.gather``: RCCL over xGMI on GPUs, gloo in the CPU tests).  This replaces the
reference's process-per-env ``VectorEnv`` + ``multiprocessing.Pipe`` pickling (vector_env.py:186-217) and the
zmq fan-out of ``BatchEnvSampler`` (rl/rollout/batch_env_sampler.py:53-97).
"""
from __future__ import annotations

import contextlib
from typing import Callable, Dict, Optional, Tuple

import torch


def shard_range(total_envs: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of the global env index range owned by `rank` (sizes differ by at most 1)."""
    base, rem = divmod(total_envs, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def rollout(engine, n_steps: int, policy: Callable[[int, torch.Tensor, torch.Tensor], Tuple[torch.Tensor, torch.Tensor]],
            first_step: bool = True) -> Dict[str, torch.Tensor]:
    """Run `n_steps` batched env-steps.  `policy(step, decisions, done) -> (actions [n,A,4], n_actions [n])` works on
    device tensors.  Returns the trajectory: decisions [T,n,8], actions [T,n,A,4], metrics [T,n,3], done [T,n]."""
    import contextlib
    st = getattr(engine, "_bound_stream", None)   # an engine bound to a side stream: the tensor ops below go to that stream too
    with (torch.cuda.stream(st) if st is not None else contextlib.nullcontext()):
        dec, met, done = engine.step() if first_step else (engine.decisions, engine.metrics, engine.done)
        D, A, M, Dn = [], [], [], []
        for t in range(n_steps):
            actions, n_actions = policy(t, dec, done)
            D.append(dec.clone())
            A.append(actions.clone())
            dec, met, done = engine.step(actions, n_actions, mask=(done == 0).to(torch.uint8))
            M.append(met.clone())
            Dn.append(done.clone())
        return {"decisions": torch.stack(D), "actions": torch.stack(A), "metrics": torch.stack(M), "done": torch.stack(Dn)}


_SHARD_SIZES: Dict[object, list] = {}   # sharding token (the caller's, identical on every rank) -> env count of every rank


class Transport:
    """What the exchanges below hand to ``torch.distributed``: the tensors themselves (device tensors go to RCCL as they
    are — no staging copy).  The functions take a `transport` so that a harness whose process group cannot carry device
    tensors supplies its own staging (tests/transport.py does, for N ranks sharing one GPU under gloo); this file holds
    only the path a multi-GPU run takes."""

    def outbound(self, t: torch.Tensor) -> torch.Tensor:
        """The tensor a send / broadcast / all_gather is posted with."""
        return t

    def landing(self, shape, like: torch.Tensor) -> torch.Tensor:
        """A receive buffer of `shape` for data that left its rank as `outbound(like)`."""
        return torch.empty(shape, dtype=like.dtype, device=like.device)

    def inbound(self, t: torch.Tensor, device: torch.device) -> torch.Tensor:
        """A received (or joined) tensor as the caller wants it, on `device`."""
        return t


_DIRECT = Transport()


def gather_to_learner(traj: Dict[str, torch.Tensor], dst: int = 0, group=None, sizes: Optional[list] = None,
                      sharding_token=None, transport: Optional[Transport] = None) -> Optional[Dict[str, torch.Tensor]]:
    """Concatenate every rank's trajectory along the env axis (dim 1 of tensors shaped [T, n_local, ...]) on rank `dst`.

    ONE grouped exchange (`batch_isend_irecv`: a single RCCL group of point-to-point transfers over xGMI on GPUs, plain
    send/recv under gloo): every rank sends each tensor exactly as it is — no padding copy, no packing copy — and `dst`
    receives each rank's piece at its true size, then joins the pieces per key (the only copy, on the learner).  Shards may
    differ in size.  The per-rank env counts come from, in this order: `sizes` (e.g. computed with `shard_range`: no
    collective at all); a previous call with the same `sharding_token` (a value EVERY rank passes identically and changes
    whenever ANY rank's shard changes — the cache is keyed on nothing a single rank could decide alone); otherwise they are
    exchanged in this call (one small all_gather, which also checks that the step count T is the same on every rank).
    Returns the dict on `dst`, None elsewhere.
    Replaces the reference's pickle-over-Pipe result collection (vector_env.py:186-217) and the zmq fan-in of
    BatchEnvSampler (rl/rollout/batch_env_sampler.py:150-190)."""
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return traj
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    tp = transport or _DIRECT
    keys = sorted(traj)
    any_t = traj[keys[0]]
    n_local, t_local = int(any_t.shape[1]), int(any_t.shape[0])
    if sizes is None and sharding_token is not None and sharding_token in _SHARD_SIZES:
        sizes = _SHARD_SIZES[sharding_token]
    if sizes is None:
        mine = tp.outbound(torch.tensor([n_local, t_local], dtype=torch.int64, device=any_t.device))
        allsz = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allsz, mine, group=group)
        got = [[int(v) for v in x.tolist()] for x in allsz]
        assert all(g[1] == t_local for g in got), f"gather_to_learner: the step count T differs across ranks: {[g[1] for g in got]}"
        sizes = [g[0] for g in got]
        if sharding_token is not None:
            _SHARD_SIZES[sharding_token] = sizes
    assert len(sizes) == world and sizes[rank] == n_local, "sizes must list every rank's env count"
    src = {k: tp.outbound(traj[k].contiguous()) for k in keys}   # (already contiguous in a rollout loop: no copy)
    dev = any_t.device
    if rank != dst:
        if n_local > 0:                              # (an empty shard sends nothing; the learner posts no receive for it)
            ops = [dist.P2POp(dist.isend, src[k], dst, group) for k in keys]
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return None
    pieces = {k: [None] * world for k in keys}
    ops = []
    for r in range(world):
        for k in keys:
            if r == rank:
                pieces[k][r] = src[k]
            else:
                shape = list(src[k].shape)
                shape[1] = sizes[r]
                pieces[k][r] = tp.landing(shape, src[k])
                if sizes[r] > 0:
                    ops.append(dist.P2POp(dist.irecv, pieces[k][r], r, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return {k: tp.inbound(torch.cat(pieces[k], dim=1), dev) for k in keys}


def gather_experiences_to_learner(exp: Dict[str, torch.Tensor], env_offset: int = 0, dst: int = 0, group=None,
                                  transport: Optional[Transport] = None) -> Optional[Dict[str, torch.Tensor]]:
    """The experiences ONE ``CimBatchSampler.sample`` / ``sample_fused`` call emitted on every rank, joined on `dst` — the
    learner-side collection of config 5 (the reference: ``BatchEnvSampler.sample`` merging its workers' results,
    maro/rl/rollout/batch_env_sampler.py:150-190).  The flat tensors over a rank's K emitted elements (state [K, D], action,
    env_action, reward, next_state, next_agent_state, terminal, env_id, tick, agent) are concatenated rank by rank, `env_id`
    shifted by the rank's `env_offset` into global env ids; with contiguous shards (``shard_range``) that is exactly the order a
    single sampler over all envs emits.  "env_metric" [n_local, 3] is joined the same way.  K differs from rank to rank and call
    to call, so the sizes are exchanged in the call (one small all_gather), then one grouped send / receive per key set as in
    ``gather_to_learner``.  Returns the dict on `dst`, None elsewhere."""
    per_elem = {k: v for k, v in exp.items() if k != "env_metric"}
    if "env_id" in per_elem and env_offset:
        per_elem["env_id"] = per_elem["env_id"] + int(env_offset)
    out = gather_to_learner({k: v.unsqueeze(0) for k, v in per_elem.items()}, dst=dst, group=group, transport=transport)
    met = gather_to_learner({"env_metric": exp["env_metric"].unsqueeze(0)}, dst=dst, group=group, transport=transport) if "env_metric" in exp else None
    if out is None:
        return None
    res = {k: v.squeeze(0) for k, v in out.items()}
    if met is not None:
        res["env_metric"] = met["env_metric"].squeeze(0)
    return res


def broadcast_policy(packed: Optional[torch.Tensor], actors=(), src: int = 0, group=None, like: Optional[torch.Tensor] = None,
                     transport: Optional[Transport] = None) -> torch.Tensor:
    """The return half of config 5's loop: the learner's refreshed networks to every sampler rank — ONE ``dist.broadcast`` of the
    packed weight blob (22 nets x ~90 k floats = ~8 MB for the CIM example; RCCL over xGMI on GPUs, gloo in the CPU tests), then
    an in-place, stream-ordered ``set_policy_state`` on each of the rank's actors (one ``FusedPerPortDQN`` per env group).
    Replaces the reference's ``policy_state`` dict pickled into every ``sample`` request and applied by each worker
    (maro/rl/rollout/batch_env_sampler.py:150-176, worker.py:56-67, env_sampler.py:37-46).

    `packed`: on `src` the new blob (``policy.pack_policy(chains)`` / ``actor.pack(chains)``, CPU or device); ignored on the
    other ranks, which receive into a buffer shaped like their first actor's weights (or `like`).  Returns the blob every rank
    now holds (on the actors' device).  With no process group (single rank) it is just the local update."""
    import torch.distributed as dist

    actors = list(actors)
    ref = actors[0].weights if actors else like
    multi = dist.is_initialized() and dist.get_world_size(group) > 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if not multi or rank == src:
        assert packed is not None, "the source rank must pass the packed policy"
        buf = packed if ref is None else packed.to(device=ref.device, dtype=torch.float32)
    else:
        assert ref is not None, "a receiving rank needs actors (or `like`) to know the blob's shape"
        buf = torch.empty_like(ref)
    buf = buf.contiguous()
    if multi:
        tp = transport or _DIRECT
        wire = tp.outbound(buf)
        dist.broadcast(wire, src=src, group=group)
        buf = tp.inbound(wire, buf.device)
    for a in actors:
        a.set_policy_state(buf)
    return buf


class PipelinedCimBatch:
    """A per-GPU env batch split into `groups` independent `CimBatchEngine`s, each on its own HIP stream.

    A step kernel ends with a tail of its slowest (ticking) waves while most of the chip is already idle; with the batch
    split into a few groups whose per-step work is issued back to back on separate streams, the other groups' kernels
    fill that tail (+30 % env-steps/s at 16384 envs).  Envs never interact, so this is pure scheduling:
    env e of the batch behaves exactly as in a single engine created with the same seeds.

        batch = PipelinedCimBatch("global_trade.22p_l0.8", 16384, groups=3, durations=1120, max_snapshots=4, specialize=True)
        batch.for_each(lambda g, eng: eng.step())                          # first step of the episode
        batch.for_each(lambda g, eng: (policy(g, eng), eng.step(a[g], n[g])))   # one rollout step, all groups
        batch.synchronize()
    """

    def __init__(self, topology, n_envs: int, groups: int = 3, seeds=None, device="cuda:0", engine_factory=None, **engine_kwargs):
        """engine_factory(topology, n, seeds=..., **engine_kwargs) -> engine: defaults to CimBatchEngine on `device`; the CPU
        suite passes the wave-emulator engine (no streams then)."""
        self.device = torch.device(device)
        groups = max(1, min(int(groups), int(n_envs)))
        self.n_envs = int(n_envs)
        self.sizes = [n_envs // groups + (1 if g < n_envs % groups else 0) for g in range(groups)]
        self.offsets = [sum(self.sizes[:g]) for g in range(groups)]
        if isinstance(seeds, str):
            assert seeds == "topology"      # every env starts from the topology's own seed, like a fresh reference Env
            seeds = None
        elif seeds is None and engine_factory is None:
            seeds = torch.arange(n_envs, dtype=torch.int64)
        if seeds is not None:
            seeds = torch.as_tensor(seeds, dtype=torch.int64)
        if engine_factory is None:
            from .engine import CimBatchEngine

            def engine_factory(topo, n, **kw):
                return CimBatchEngine(topo, n, device=self.device, **kw)
        self.engines = [engine_factory(topology, self.sizes[g],
                                       seeds=None if seeds is None else seeds[self.offsets[g]:self.offsets[g] + self.sizes[g]], **engine_kwargs)
                        for g in range(groups)]
        on_gpu = self.device.type == "cuda" and hasattr(self.engines[0], "use_stream")
        self.streams = [torch.cuda.Stream(device=self.device) if on_gpu else None for _ in range(groups)]
        for eng, st in zip(self.engines, self.streams):
            if st is not None:
                eng.use_stream(st)   # engine calls go to the group's stream even outside for_each() (see CimBatchEngine.use_stream
                                     # for the stream discipline: read outputs inside for_each() or after synchronize())
        if on_gpu:
            torch.cuda.synchronize(self.device)

    def __len__(self) -> int:
        return len(self.engines)

    def for_each(self, fn: Callable) -> list:
        """Run fn(group_index, engine) for every group with that group's stream current; returns the results."""
        out = []
        for g, (eng, st) in enumerate(zip(self.engines, self.streams)):
            if st is None:
                out.append(fn(g, eng))
                continue
            with torch.cuda.stream(st):
                out.append(fn(g, eng))
        return out

    def synchronize(self) -> None:
        for st in self.streams:
            if st is not None:
                st.synchronize()

    # ------------------------------------------------------------------ the CimBatchEngine surface over the whole batch
    # What GpuVectorEnv needs from "its engine" (maro_amd/cim/vector_env.py), so that the object API — VectorEnv.step / reset /
    # snapshot_list of maro/vector_env/vector_env.py:95-217 — can run on the stream-pipelined groups: whole-batch inputs are
    # split by env range, every group's call is issued on its own stream before any result is read, outputs are concatenated in
    # env order.
    topo = property(lambda self: self.engines[0].topo)
    max_actions = property(lambda self: self.engines[0].max_actions)
    start_tick = property(lambda self: self.engines[0].start_tick)
    durations = property(lambda self: self.engines[0].durations)
    snapshot_resolution = property(lambda self: self.engines[0].snapshot_resolution)
    decision_mode = property(lambda self: self.engines[0].decision_mode)
    ticks = property(lambda self: self.cat("ticks"))
    status = property(lambda self: self.cat("status"))
    ring_fi = property(lambda self: self.cat("ring_fi"))
    layout = property(lambda self: self.engines[0].layout)   # per-env dimensions (ring slots, frame words ...) are the same in every group

    def _rows(self, x, g: int):
        """Group g's env rows of a whole-batch input (None stays None)."""
        return None if x is None else x[self.offsets[g]:self.offsets[g] + self.sizes[g]]

    def _after_caller(self, *inputs) -> None:
        """Stream discipline of the whole-batch calls: a DEVICE tensor the caller produced on torch's current stream is sliced and
        handed to every group's engine, whose kernels run on the group's side stream — so each side stream first waits for the
        caller's stream (one event wait per group, only when a device tensor is actually passed; host inputs are copied on the
        group's stream by the engine itself)."""
        if any(isinstance(x, torch.Tensor) and x.is_cuda for x in inputs):
            cur = torch.cuda.current_stream(self.device)
            for st in self.streams:
                if st is not None:
                    st.wait_stream(cur)

    def step(self, actions=None, n_actions=None, mask=None, n_answered=None):
        """Whole-batch step.  Device-tensor inputs may come straight from the caller's current stream (see _after_caller); the
        returned tensors are complete (cat() synchronises the groups)."""
        self._after_caller(actions, n_actions, mask, n_answered)
        if actions is not None and not isinstance(actions, torch.Tensor):
            import numpy as np
            actions = np.asarray(actions).reshape(self.n_envs, self.max_actions, -1)
        elif actions is not None:
            actions = actions.reshape(self.n_envs, self.max_actions, -1)

        def one(g, eng):
            if n_answered is None:
                return eng.step(self._rows(actions, g), self._rows(n_actions, g), self._rows(mask, g))
            return eng.step(self._rows(actions, g), self._rows(n_actions, g), self._rows(mask, g), n_answered=self._rows(n_answered, g))
        self.for_each(one)
        return self.cat("decisions"), self.cat("metrics"), self.cat("done")

    def reset(self, seed_cmd=None, mask=None) -> None:
        self._after_caller(seed_cmd, mask)
        self.for_each(lambda g, eng: eng.reset(self._rows(seed_cmd, g), self._rows(mask, g)))

    def query(self, node: str, ticks, nodes, attrs, out=None) -> torch.Tensor:
        """As CimBatchEngine.query; per-env `ticks` [n_envs, nt] / `nodes` [n_envs, nn] are split by env range, shared ones
        ([nt] / [nn]) go to every group."""
        def part(x, g):
            return self._rows(x, g) if getattr(x, "ndim", 1) == 2 else x
        self._after_caller(ticks, nodes)
        res = self.for_each(lambda g, eng: eng.query(node, part(ticks, g), part(nodes, g), attrs))
        self.synchronize()
        return torch.cat(res, dim=0)

    def clear_status_bits(self, envs, bits: int) -> None:
        """status[e] &= ~bits for the listed envs (the object API acknowledges MRX_ENV_INVALID_ACTION this way)."""
        for e in envs:
            g = max(i for i, o in enumerate(self.offsets) if o <= e)
            with torch.cuda.stream(self.streams[g]) if self.streams[g] is not None else contextlib.nullcontext():
                self.engines[g].status[e - self.offsets[g]] &= ~int(bits)

    def cat(self, name: str) -> torch.Tensor:
        """Concatenate a per-engine tensor attribute (decisions, metrics, done, ticks, status ...) in env order."""
        self.synchronize()
        return torch.cat([getattr(e, name) for e in self.engines], dim=0)
