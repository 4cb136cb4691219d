"""Batched, on-device observation and reward shaping for the CIM RL example.

Mirrors ``examples/cim/rl/env_sampler.py:15-31`` (state) and ``:65-80`` (delayed reward) with the shaping
constants of ``examples/cim/rl/config.py:13-29``, for every env of a ``CimBatchEngine`` at once: three snapshot
queries and a handful of tensor ops, no per-env Python loop and no host round trip.
"""
from __future__ import annotations

import ctypes
import os
from typing import Callable, Dict, Optional, Sequence

import torch

PORT_ATTRIBUTES = ["empty", "full", "on_shipper", "on_consignee", "booking", "shortage", "fulfillment"]
VESSEL_ATTRIBUTES = ["empty", "full", "remaining_space"]


class CimBatchSampler:
    def __init__(self, engine, look_back: int = 7, time_window: int = 99, fulfillment_factor: float = 1.0,
                 shortage_factor: float = 1.0, time_decay: float = 0.97, port_attributes: Sequence[str] = PORT_ATTRIBUTES,
                 vessel_attributes: Sequence[str] = VESSEL_ATTRIBUTES):
        self.eng = engine
        self.look_back, self.time_window = look_back, time_window
        self.ff, self.sf = fulfillment_factor, shortage_factor
        self.reward_eval_delay = time_window   # examples/cim/rl/rl_component_bundle: reward_eval_delay = reward_shaping_conf["time_window"]
        self.port_attributes, self.vessel_attributes = list(port_attributes), list(vessel_attributes)
        dev = engine.decisions.device
        self._back = torch.arange(look_back - 1, dtype=torch.int32, device=dev)          # range(look_back - 1)
        self._ahead = torch.arange(1, time_window + 1, dtype=torch.int32, device=dev)    # tick + 1 ... tick + window
        self._decay = torch.tensor([time_decay ** i for i in range(time_window)], dtype=torch.float64, device=dev)
        self.state_dim = (look_back - 1) * (1 + engine.topo.future_stop_number) * len(self.port_attributes) + len(
            self.vessel_attributes)

    def state(self, decisions: Optional[torch.Tensor] = None) -> torch.Tensor:
        """float64 [n_envs, state_dim] for the pending decision of every env (rows of finished envs are meaningless).
        env_sampler.py:21-31: ports = [port] + future_stop_list of the vessel; ticks = max(0, tick - rt)."""
        d = self.eng.decisions if decisions is None else decisions
        n = d.shape[0]
        tick, port, vessel = d[:, 0:1], d[:, 1:2], d[:, 2:3]
        fut = self.eng.query("vessels", tick, vessel, ["future_stop_list"]).view(n, -1).to(torch.int32)
        nodes = torch.cat([port.to(torch.int32), fut], dim=1).contiguous()
        ticks = torch.clamp(tick - self._back[None, :], min=0).to(torch.int32).contiguous()
        ps = self.eng.query("ports", ticks, nodes, self.port_attributes).view(n, -1)
        vs = self.eng.query("vessels", tick, vessel, self.vessel_attributes).view(n, -1)
        return torch.cat([ps, vs], dim=1)

    def reward(self, tick: torch.Tensor, port: torch.Tensor) -> torch.Tensor:
        """float32 [n_envs]: sum_i decay^i * (ff * fulfillment - sf * shortage) of `port` over ticks tick+1 .. tick+window
        (env_sampler.py:65-80).  Frames the ring does not hold contribute zeros, like the reference's padding."""
        n = tick.shape[0]
        ticks = (tick.view(n, 1).to(torch.int32) + self._ahead[None, :]).contiguous()
        nodes = port.view(n, 1).to(torch.int32).contiguous()
        q = self.eng.query("ports", ticks, nodes, ["fulfillment", "shortage"]).view(n, self.time_window, 2)
        r = self.ff * (q[:, :, 0] @ self._decay) - self.sf * (q[:, :, 1] @ self._decay)
        return r.to(torch.float32)

    # ------------------------------------------------------------------ AbsEnvSampler.sample, batched
    #: slots per env of a fresh transition cache (a power of two; it doubles when an env's live elements no longer fit)
    INITIAL_CACHE_SLOTS = 256

    def _cache_alloc(self, cap: int) -> None:
        """The per-env transition cache is a RING of `cap` slots (a power of two): element number q of an env (`_head` counts
        the elements it has appended, `_tail` the ones it has emitted or dropped) lives in slot q & (cap - 1), so emitting
        the oldest elements moves `_tail` and nothing else (a linear cache had to shift everything behind them: at 16384 envs
        x 512 slots that was 17 GB per emission)."""
        cap = 1 << max(1, int(cap - 1).bit_length())
        n, D, dev = self.eng.n_envs, self.state_dim, self.eng.decisions.device
        old = getattr(self, "_c", None)
        c = dict(tick=torch.zeros((n, cap), dtype=torch.int32, device=dev), agent=torch.zeros((n, cap), dtype=torch.int64, device=dev),
                 state=torch.zeros((n, cap, D), dtype=self.state_dtype, device=dev), action=torch.zeros((n, cap), dtype=torch.int64, device=dev),
                 env_action=torch.zeros((n, cap, 4), dtype=torch.int32, device=dev),
                 next_state=torch.zeros((n, cap, D), dtype=self.state_dtype, device=dev),
                 next_agent_state=torch.zeros((n, cap, D), dtype=self.state_dtype, device=dev),
                 terminal=torch.zeros((n, cap), dtype=torch.bool, device=dev))
        if old is not None:      # growth: every live element moves to its slot in the larger ring
            k = old["tick"].shape[1]
            sl = torch.arange(k, device=dev)[None, :]
            seq = self._tail[:, None] + ((sl - self._tail[:, None]) & (k - 1))     # the element an old slot would hold
            er, es = torch.nonzero(seq < self._head[:, None], as_tuple=True)
            ns = seq[er, es] & (cap - 1)
            for key in c:
                c[key][er, ns] = old[key][er, es]
            if hasattr(self, "_pj"):   # the slot of every env's newest element in the larger ring (device-resident loop)
                self._pj.copy_((self._head - 1).clamp(min=0) & (cap - 1))
        self._c, self._cap = c, cap

    def _sample_init(self, state_dtype) -> None:
        eng = self.eng
        n, dev = eng.n_envs, eng.decisions.device
        assert eng.start_tick == 0 and eng.snapshot_resolution == 1, "the CIM example's shaping indexes snapshots by tick"
        self.state_dtype = state_dtype
        self._c = None
        self._cache_alloc(self.INITIAL_CACHE_SLOTS)
        self._head = torch.zeros(n, dtype=torch.int64, device=dev)      # elements appended so far (the next element's number)
        self._tail = torch.zeros(n, dtype=torch.int64, device=dev)      # number of the oldest element still cached
        self._last = torch.full((n, eng.layout.n_ports), -1, dtype=torch.int64, device=dev)   # _agent_last_index (element numbers)
        self._eoe = torch.ones(n, dtype=torch.bool, device=dev)                               # _end_of_episode
        self._episodes = 0                                   # reset_envs calls so far (diagnostics)
        self._steps_env = torch.zeros(n, dtype=torch.int64, device=dev)   # interactions each env has performed (sample_fused)
        self._ep_env = torch.zeros(n, dtype=torch.int64, device='cpu')   # episodes each env has started: what its seed depends on
        self._cur_state = torch.zeros((n, self.state_dim), dtype=state_dtype, device=dev)
        self._pj = torch.zeros(n, dtype=torch.int64, device=dev)       # slot the env's previous interaction wrote (device-resident loop)
        self._pa = torch.zeros(n, dtype=torch.uint8, device=dev)       # ... and whether that element still waits for its next state
        self._max_cached, self._any_eoe = 0, True                     # host-side: bound on any env's cached elements; may an env need a reset
        self._dev_pending = False                                      # a device-resident call left newest elements waiting for their next state
        # per-attribute retention: the whole episode of (fulfillment, shortage) per port, written by the step kernel at every
        # snapshot — the delayed reward reads up to time_window ticks ahead of decisions that may be an episode old
        self._hist = eng.set_port_history(["fulfillment", "shortage"])
        self._ports = torch.arange(eng.layout.n_ports, dtype=torch.int32, device=dev)

    def _finalize_and_emit(self, envs: torch.Tensor, out: Dict[str, list], kernel: bool = False) -> None:
        """`_append_cache_element(None)` + the emission loop of AbsEnvSampler.sample (rl/rollout/env_sampler.py:404-410,
        514-530) for the envs in the bool mask `envs`."""
        eng, c = self.eng, self._c
        n, cap = c["tick"].shape
        dev = c["tick"].device
        ring = cap - 1
        rows = torch.nonzero(envs).view(-1)
        if rows.numel() == 0:
            return
        # last element of every agent: terminal = end_of_episode, next agent state = its own state
        li = self._last[rows]                                   # [m, P]
        has = li >= 0
        r_idx = rows[:, None].expand_as(li)[has]
        j_idx = li[has] & ring
        c["terminal"][r_idx, j_idx] = self._eoe[r_idx]
        c["next_agent_state"][r_idx, j_idx] = c["state"][r_idx, j_idx]
        # elements old enough for their reward window: tick <= env.tick - reward_eval_delay (ticks are non-decreasing: a prefix)
        tick_now = eng.ticks.to(torch.int64)
        bound = tick_now[rows] - self.reward_eval_delay
        pos = torch.arange(cap, device=dev)[None, :]
        tail = self._tail[rows]
        slot_of = (tail[:, None] + pos) & ring                            # slot of an env's j-th oldest element   [m, cap]
        tick_j = torch.gather(c["tick"][rows], 1, slot_of).to(torch.int64)
        emit = (pos < (self._head[rows] - tail)[:, None]) & (tick_j <= bound[:, None])
        n_emit = emit.sum(dim=1)
        if int(n_emit.sum()) > 0:
            # the frame of the tick an env is paused at is its live frame (pre-decision snapshot, core.py:345): its retention
            # row is only written when the tick completes, so it is filled in from a snapshot query here
            paused = rows[~self._eoe[rows]]
            if paused.numel() > 0:
                live = eng.query("ports", tick_now.to(torch.int32).view(n, 1), self._ports, ["fulfillment", "shortage"]).view(n, -1, 2)
                self._hist[paused, tick_now[paused]] = live[paused].permute(0, 2, 1).to(torch.int32)
            frames = self._hist.shape[1]
            emit_fn = getattr(getattr(eng, "_L", None), "mrx_cim_sampler_emit", None) if kernel and dev.type == "cuda" else None
            if emit_fn is not None:
                # ONE launch (mrx_cim_sampler_emit): an env's oldest n_emit elements copied out in age order, rewards evaluated
                # on the way.  The tensor-op sequence of the other branch is its specification (rewards: same float64 sums in
                # another order, so equal up to the last float32 bit).
                from .. import _lib
                K = int(n_emit.sum())
                off = torch.cumsum(n_emit, 0) - n_emit
                D = self.state_dim
                o = dict(state=torch.empty((K, D), dtype=self.state_dtype, device=dev), action=torch.empty(K, dtype=torch.int64, device=dev),
                         env_action=torch.empty((K, 4), dtype=torch.int32, device=dev), reward=torch.empty(K, dtype=torch.float32, device=dev),
                         next_state=torch.empty((K, D), dtype=self.state_dtype, device=dev),
                         next_agent_state=torch.empty((K, D), dtype=self.state_dtype, device=dev), terminal=torch.empty(K, dtype=torch.bool, device=dev),
                         env_id=torch.empty(K, dtype=torch.int32, device=dev), tick=torch.empty(K, dtype=torch.int32, device=dev),
                         agent=torch.empty(K, dtype=torch.int32, device=dev))
                rows_c, tail_c, n_emit_c = rows.contiguous(), tail.contiguous(), n_emit.contiguous()
                assert self._hist.is_contiguous() and self._hist.shape[2] == 2
                _lib.check(emit_fn(int(rows_c.numel()), eng.layout.n_ports, D, cap, frames, self.time_window, 1 if self.state_dtype == torch.float64 else 0,
                                   float(self.ff), float(self.sf), self._decay.data_ptr(), rows_c.data_ptr(), tail_c.data_ptr(), n_emit_c.data_ptr(),
                                   off.data_ptr(), self._hist.data_ptr(), c["tick"].data_ptr(), c["agent"].data_ptr(), c["state"].data_ptr(),
                                   c["action"].data_ptr(), c["env_action"].data_ptr(), c["terminal"].data_ptr(), c["next_state"].data_ptr(),
                                   c["next_agent_state"].data_ptr(), o["state"].data_ptr(), o["action"].data_ptr(), o["env_action"].data_ptr(),
                                   o["reward"].data_ptr(), o["next_state"].data_ptr(), o["next_agent_state"].data_ptr(), o["terminal"].data_ptr(),
                                   o["env_id"].data_ptr(), o["tick"].data_ptr(), o["agent"].data_ptr(),
                                   dev.index if dev.index is not None else torch.cuda.current_device(), eng._stream()), "mrx_cim_sampler_emit")
                self._keep_emit = (rows_c, tail_c, n_emit_c, off)      # (inputs of an asynchronous launch)
                for key, v in o.items():
                    out[key].append(v)
            else:
                er, ej = torch.nonzero(emit, as_tuple=True)            # ordered by env, then by age: the reference's emission order
                e_env = rows[er]
                ej = slot_of[er, ej]
                tick = c["tick"][e_env, ej].to(torch.int64)
                agent = c["agent"][e_env, ej]
                t_idx = tick[:, None] + self._ahead.to(torch.int64)[None, :]                 # tick + 1 .. tick + window
                ok = t_idx < frames                                                          # beyond the episode: zeros (snapshot padding)
                vals = self._hist[e_env[:, None], t_idx.clamp(max=frames - 1), :, agent[:, None]].to(torch.float64) * ok[:, :, None]   # [K, window, 2]
                reward = (self.ff * (vals[:, :, 0] @ self._decay) - self.sf * (vals[:, :, 1] @ self._decay)).to(torch.float32)
                out["env_id"].append(e_env.to(torch.int32)); out["tick"].append(tick.to(torch.int32)); out["agent"].append(agent.to(torch.int32))
                out["reward"].append(reward)
                for key in ("state", "action", "env_action", "next_state", "next_agent_state", "terminal"):
                    out[key].append(c[key][e_env, ej])
            # pop the emitted prefix: the ring's tail moves on; an agent whose last element went out has none
            new_tail = tail + n_emit
            self._tail[rows] = new_tail
            li = self._last[rows]
            self._last[rows] = torch.where(li >= new_tail[:, None], li, torch.full_like(li, -1))

    def sample(self, policy: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], num_steps: Optional[int] = None,
               seeds: Optional[Callable[[torch.Tensor], torch.Tensor]] = None, state_dtype: torch.dtype = torch.float32) -> Dict[str, torch.Tensor]:
        """``AbsEnvSampler.sample(num_steps)`` (rl/rollout/env_sampler.py:438-537) with the CIM example's shaping
        (examples/cim/rl/env_sampler.py:15-80) for every env of the engine at once, on device tensors.

        `policy(states [n, state_dim], decisions [n, 8]) -> model actions int64 [n]` (indices into the example's action
        space; rows of envs without a pending decision are ignored).  Every env performs `num_steps` interactions (None:
        until the end of ITS episode); an env whose episode ends inside the call is reset (`seeds(episode_index: CPU int64 tensor [n]) -> int64
        [n]` — see `_seed_fn`: explicit seeds from every env's OWN episode count, or the reference's seed re-draw) and goes on, exactly like the reference loop.  Transitions
        are emitted once they are `reward_eval_delay` (= time_window) ticks old — the delayed reward is evaluated after
        the loop on the per-attribute retention rows (mrx_cim_set_port_history), so the snapshot ring can stay a few
        frames deep (look_back) — and younger ones stay in the per-env cache for the next call, with the per-agent next
        state / terminal bookkeeping of ``_append_cache_element``.

        Returns flat tensors over the K emitted experiences, ordered by emission: state [K, D], action int64 [K]
        (model action), env_action int32 [K, 4], reward float32 [K], next_state [K, D], next_agent_state [K, D],
        terminal bool [K], env_id int32 [K], tick int32 [K], agent int32 [K] (the deciding port).  Plus "env_metric"
        int64 [n, 3] (`_post_step`)."""
        from .engine import SEED_REDRAW
        from .policy import translate_actions
        eng = self.eng
        n, dev = eng.n_envs, eng.decisions.device
        seeds = _seed_fn(seeds, n)
        if not hasattr(self, "_c") or self.state_dtype != state_dtype:
            self._sample_init(state_dtype)
        self._drain_device_pending()
        c = self._c
        out = {k: [] for k in ("state", "action", "env_action", "reward", "next_state", "next_agent_state", "terminal", "env_id", "tick", "agent")}
        acts = torch.zeros((n, eng.max_actions, 4), dtype=torch.int32, device=dev)
        nact = torch.zeros(n, dtype=torch.int32, device=dev)
        steps_to_go = num_steps

        def reset_envs(mask: torch.Tensor) -> None:   # AbsEnvSampler._reset for the envs in `mask`
            # `seeds` sees every env's OWN episode index (the reference runs one sampler loop per env: episode k of an env gets
            # seed k of that env, whatever the rest of the batch is doing); only the masked envs' entries are used
            cmd = seeds(self._ep_env.clone()).to(torch.int64) if seeds is not None else torch.full((n,), SEED_REDRAW, dtype=torch.int64)
            self._episodes += 1
            self._ep_env += mask.cpu().to(torch.int64)
            eng.reset(cmd, mask.to(torch.uint8))
            rows = torch.nonzero(mask).view(-1)
            self._hist[rows] = 0
            self._tail[rows] = self._head[rows]
            self._last[rows] = -1
            eng.step(mask=mask.to(torch.uint8))          # _step(None): the first decision event
            self._eoe = torch.where(mask, eng.done.to(torch.bool), self._eoe)
            st = self.state().to(self.state_dtype)
            self._cur_state = torch.where(mask[:, None], st, self._cur_state)

        if bool(self._eoe.any()):
            reset_envs(self._eoe.clone())
        while True:
            if num_steps is None:
                if bool(self._eoe.all()):
                    break
            else:
                if steps_to_go == 0:
                    break
                if bool(self._eoe.any()):                # the outer loop of the reference: finish that episode's batch, reset, go on
                    ended = self._eoe.clone()
                    self._finalize_and_emit(ended, out)
                    reset_envs(ended)
            active = ~self._eoe
            dec = eng.decisions.clone()
            state = self._cur_state
            model_action = policy(state, dec).to(torch.int64)
            translate_actions(model_action, dec, state[:, -1].to(torch.float64), dec[:, 5], out=acts)   # vessel remaining_space = last state entry
            nact[:] = active.to(torch.int32)
            if int((self._head - self._tail).max()) >= self._cap:
                self._cache_alloc(2 * self._cap)
                c = self._c
            rows = torch.nonzero(active).view(-1)
            seq = self._head[rows]
            j = seq & (self._cap - 1)
            agent = dec[rows, 1].to(torch.int64)
            c["tick"][rows, j] = dec[rows, 0]
            c["agent"][rows, j] = agent
            c["state"][rows, j] = state[rows]
            c["action"][rows, j] = model_action[rows]
            c["env_action"][rows, j] = acts[rows, 0]
            c["terminal"][rows, j] = False
            prev = self._last[rows, agent]               # this agent's previous element gets its next agent state
            hp = prev >= 0
            pslot = prev[hp] & (self._cap - 1)
            c["next_agent_state"][rows[hp], pslot] = state[rows[hp]]
            c["terminal"][rows[hp], pslot] = False
            self._last[rows, agent] = seq
            self._head[rows] += 1
            eng.step(acts, nact, mask=active.to(torch.uint8))
            self._eoe = torch.where(active, eng.done.to(torch.bool), self._eoe)
            st = self.state().to(self.state_dtype)
            self._cur_state = torch.where((active & ~self._eoe)[:, None], st, self._cur_state)
            c["next_state"][rows, j] = self._cur_state[rows]
            if steps_to_go is not None:
                steps_to_go -= 1
        self._finalize_and_emit(torch.ones(n, dtype=torch.bool, device=dev), out)
        res = {k: (torch.cat(v) if v else torch.zeros((0,) + tuple(c[k].shape[2:]) if k in c else (0,), dtype=(c[k].dtype if k in c else torch.int32), device=dev))
               for k, v in out.items()}
        if not out["reward"]:
            res["reward"] = torch.zeros(0, dtype=torch.float32, device=dev)
        res["env_metric"] = eng.metrics.clone()
        self._after_per_step_call()
        return res

    def eval(self, policy, num_episodes: int = 1, seeds: Optional[Callable[[torch.Tensor], torch.Tensor]] = None, done_every: int = 32) -> dict:
        """``AbsEnvSampler.eval(num_episodes)`` (rl/rollout/env_sampler.py:561-611) for every env of the engine at once: per
        episode a reset, then the policy's EXPLOITING actions (greedy: ``agent_wrapper.exploit()``) until every env's episode is
        over.  The reference builds cache elements here only to feed ``_post_eval_step``, which the CIM example leaves empty
        (examples/cim/rl/env_sampler.py:85-86), so what an episode leaves behind is ``info["env_metric"]`` (``_post_step``).

        `policy`: a ``FusedPerPortDQN`` of this engine (the on-device forward; its epsilon is ignored here) or a callable
        ``(states [n, state_dim], decisions [n, 8]) -> model actions``.  `seeds` as in ``sample`` (default: the reference's
        seed re-draw).  `done_every`: interactions enqueued between two reads of the done flags.

        Returns ``{"info": [{"env_metric": int64 [n_envs, 3]}, ...]}`` — one entry per episode, one row per env (the reference
        evaluates on its separate ``test_env``; here the sampler's own engine IS the test batch: keep one sampler over a test
        engine for ``eval`` and another for ``sample``, as the reference keeps two envs — a ``sample`` after an ``eval`` on the
        same sampler starts from fresh episodes)."""
        from .engine import SEED_REDRAW
        from .policy import translate_actions
        eng = self.eng
        n, dev = eng.n_envs, eng.decisions.device
        seeds = _seed_fn(seeds, n)
        fused = hasattr(policy, "act") and hasattr(policy, "_m")
        acts = torch.zeros((n, eng.max_actions, 4), dtype=torch.int32, device=dev)
        nact = torch.zeros(n, dtype=torch.int32, device=dev)
        if hasattr(self, "_c"):
            self._drain_device_pending()
        ep_env = torch.zeros(n, dtype=torch.int64)
        info_list = []
        eps_saved = None
        if fused:
            eps_saved, policy._m.epsilon = policy._m.epsilon, 0.0   # exploit()
        try:
            for _ in range(int(num_episodes)):
                cmd = seeds(ep_env.clone()).to(torch.int64) if seeds is not None else torch.full((n,), SEED_REDRAW, dtype=torch.int64)
                ep_env += 1
                eng.reset(cmd)
                eng.step()                                   # _step(None): the first decision event
                # a finished env answers further steps with (None, None, True) — metrics zero — so its metrics are kept from the
                # step that ended its episode (no host sync: two selects per interaction)
                final = torch.zeros_like(eng.metrics)
                seen = torch.zeros(n, dtype=torch.bool, device=dev)

                def keep_final():
                    over = eng.done.to(torch.bool)
                    final.copy_(torch.where((over & ~seen)[:, None], eng.metrics, final))
                    seen.logical_or_(over)

                keep_final()
                while not bool(eng.done.all()):
                    for _k in range(max(1, int(done_every))):
                        if fused:
                            policy.act(acts, nact)           # finished envs have no pending decision: n_actions 0
                        else:
                            dec = eng.decisions.clone()
                            state = self.state(dec)
                            model_action = policy(state.to(torch.float32), dec).to(torch.int64)
                            translate_actions(model_action, dec, state[:, -1].to(torch.float64), dec[:, 5], out=acts)
                            nact[:] = (~eng.done.to(torch.bool)).to(torch.int32)
                        eng.step(acts, nact)                 # (finished envs just report `done` again)
                        keep_final()
                info_list.append({"env_metric": final})
        finally:
            if fused:
                policy._m.epsilon = eps_saved
        if hasattr(self, "_c"):                              # the engine's episodes are over: the next sample() starts afresh
            self._sample_init(self.state_dtype)
        return {"info": info_list}

    def sample_fused(self, actor, num_steps: Optional[int], seeds: Optional[Callable[[torch.Tensor], torch.Tensor]] = None, reset_every: int = 1,
                     state_dtype: torch.dtype = torch.float32) -> Dict[str, torch.Tensor]:
        """``sample_fused_steps`` run to the end (see there); returns its result."""
        gen = self.sample_fused_steps(actor, num_steps, seeds=seeds, reset_every=reset_every, state_dtype=state_dtype)
        while True:
            try:
                next(gen)
            except StopIteration as stop:
                return stop.value

    def sample_fused_steps(self, actor, num_steps: Optional[int], seeds: Optional[Callable[[torch.Tensor], torch.Tensor]] = None, reset_every: int = 1,
                           state_dtype: torch.dtype = torch.float32):
        """A GENERATOR over the interactions of one call (it yields after each step's work has been enqueued and returns the
        result dict through StopIteration) — so that the samplers of several env groups, each on its own HIP stream, can be
        advanced in turn and their kernels overlap (``sample_fused_groups``); ``sample_fused`` simply exhausts it.

        ``sample(num_steps)`` with the per-step work fused and the step path free of host synchronisation — the loop
        ``bench.py --policy dqn --collect`` times (maro.rl's EnvSampler with on-device inference).

        `actor.act(actions, n_actions, decisions=, state=, choice=)` answers every pending decision in ONE call: it writes the
        env actions `step` reads, the sampler state of each deciding env into `state` [n, state_dim] and the model action
        into `choice` int32 [n] — ``FusedPerPortDQN`` (mrx_cim_dqn_act: state gather + per-port DQN + translation, two
        launches) instead of three snapshot queries, a policy call and the translation.  The transition cache is updated with
        masked full-batch tensor ops (no ``nonzero`` / ``bool()`` / ``int()`` per step); the next state of an element is
        filled in by the NEXT step's gather (one extra state evaluation after the loop).

        `seeds`: as in ``sample`` — called with a CPU int64 tensor [n] of every env's own episode count (`_seed_fn`).

        `reset_every` = K: envs whose episode ended are finalised, emitted and reset at every K-th step only (and at the
        start of a call); in between they sit the steps out.  K = 1 is ``sample`` exactly (every env performs `num_steps`
        interactions; one flag read per step); K > 1 trades that alignment for a sync-free step path — each env's own stream
        of experiences is unchanged, an env just contributes fewer than `num_steps` steps to a call in which it rolled over."""
        from .engine import SEED_REDRAW
        eng = self.eng
        n, dev = eng.n_envs, eng.decisions.device
        seeds = _seed_fn(seeds, n)
        if eng.decisions.is_cuda and num_steps is not None and hasattr(actor, "collect_steps") and state_dtype in (torch.float32, torch.float64) \
                and 8 * eng.layout.n_ports * self.time_window <= 48 * 1024 and os.environ.get("MRX_SAMPLER_V2", "1") != "0":
            # (the emission kernel stages one reward window of an env's history rows in 48 KB of LDS: 2 x n_ports words per tick)
            # the device-resident form of this loop (same results): one C call per segment of interactions, the end of the call
            # (finalisation, emission) in three launches and one read-back
            return (yield from self._sample_fused_device(actor, num_steps, seeds, reset_every, state_dtype))
        # An engine bound to a side stream (CimBatchEngine.use_stream): the sampler's own tensor ops go to that stream too.  The
        # stream is switched per SEGMENT, never across a `yield` (another group's generator runs in between).
        bound = getattr(eng, "_bound_stream", None)

        def enter():
            if bound is None:
                return None
            prev = torch.cuda.current_stream(bound.device)
            torch.cuda.set_stream(bound)
            return prev

        def leave(prev) -> None:
            if prev is not None:
                torch.cuda.set_stream(prev)

        def reset_envs(mask: torch.Tensor) -> None:   # AbsEnvSampler._reset for the envs in `mask` (a sync point: rare)
            cmd = seeds(self._ep_env.clone()).to(torch.int64) if seeds is not None else torch.full((n,), SEED_REDRAW, dtype=torch.int64)
            self._episodes += 1
            self._ep_env += mask.cpu().to(torch.int64)
            eng.reset(cmd, mask.to(torch.uint8))
            self._hist[mask] = 0
            self._tail.copy_(torch.where(mask, self._head, self._tail))
            self._last[mask] = -1
            eng.step(mask=mask.to(torch.uint8))          # _step(None): the first decision event
            self._eoe.copy_(torch.where(mask, eng.done.to(torch.bool), self._eoe))

        def roll_over() -> None:
            if bool(self._eoe.any()):
                ended = self._eoe.clone()
                self._finalize_and_emit(ended, out, kernel=fused_kernels)
                reset_envs(ended)

        fused_kernels = bool(eng.decisions.is_cuda and num_steps is not None)   # the record / emit kernels (else: their tensor-op specification)
        tok = enter()
        try:
            if not hasattr(self, "_c") or self.state_dtype != state_dtype:
                self._sample_init(state_dtype)
            self._drain_device_pending()
            if num_steps is not None:
                need = int((self._head - self._tail).max()) + num_steps + 1      # one read per CALL: the cache never has to grow inside the loop
                if need > self._cap:
                    self._cache_alloc(max(need, 2 * self._cap))
            c = self._c
            out = {k: [] for k in ("state", "action", "env_action", "reward", "next_state", "next_agent_state", "terminal", "env_id", "tick", "agent")}
            acts = torch.zeros((n, eng.max_actions, 4), dtype=torch.int32, device=dev)
            nact = torch.zeros(n, dtype=torch.int32, device=dev)
            st_buf = torch.zeros((n, self.state_dim), dtype=torch.float32, device=dev)
            ch_buf = torch.zeros(n, dtype=torch.int32, device=dev)
            ar = torch.arange(n, device=dev)
            if bool(self._eoe.any()):
                reset_envs(self._eoe.clone())
            pj_t = torch.zeros(n, dtype=torch.int64, device=dev)
            pa_t = torch.zeros(n, dtype=torch.uint8, device=dev)
        finally:
            leave(tok)
        prev_j = None          # cache slot each env wrote in the previous step (its next_state is this step's state)
        prev_active = None
        # On the GPU the whole per-step cache update is ONE kernel (mrx_cim_sampler_record); the tensor-op sequence below is its
        # specification (and what runs on the CPU emulator, pinned by the reference sampler's goldens)
        rec = getattr(getattr(eng, "_L", None), "mrx_cim_sampler_record", None) if eng.decisions.is_cuda and num_steps is not None else None
        if rec is not None:
            from .. import _lib
            f64 = 1 if self.state_dtype == torch.float64 else 0
            assert self.state_dtype in (torch.float32, torch.float64)
            dev_index = dev.index if dev.index is not None else torch.cuda.current_device()
            first = 1
            for k in range(num_steps):
                if k > 0 and k % max(1, reset_every) == 0:
                    tok = enter()
                    try:
                        torch.logical_or(self._eoe, eng.done, out=self._eoe)   # (the record kernel folds this in; here the flags are READ)
                        if not first:
                            self._fill_next_state(pj_t, pa_t.to(torch.bool), ar)
                            first = 1
                        roll_over()
                        c = self._c
                    finally:
                        leave(tok)
                # the step itself: three C-ABI calls on the engine's stream, no tensor op, no host synchronisation.  eoe |= done
                # of the previous step happens inside the record kernel.
                actor.act(acts, nact, decisions=eng.decisions, state=st_buf, choice=ch_buf)
                _lib.check(rec(n, eng.layout.n_ports, self.state_dim, self._cap, eng.max_actions, f64, first, eng.decisions.data_ptr(), st_buf.data_ptr(),
                               ch_buf.data_ptr(), acts.data_ptr(), nact.data_ptr(), self._eoe.data_ptr(), eng.done.data_ptr(), self._head.data_ptr(),
                               self._last.data_ptr(), pj_t.data_ptr(), pa_t.data_ptr(), c["tick"].data_ptr(), c["agent"].data_ptr(), c["state"].data_ptr(),
                               c["action"].data_ptr(), c["env_action"].data_ptr(), c["terminal"].data_ptr(), c["next_state"].data_ptr(),
                               c["next_agent_state"].data_ptr(), self._steps_env.data_ptr(), dev_index, eng._stream()),
                           "mrx_cim_sampler_record")
                first = 0
                eng.step(acts, nact)                       # (finished envs just report `done` again: no mask needed)
                yield k
            tok = enter()
            try:
                torch.logical_or(self._eoe, eng.done, out=self._eoe)
                if not first:
                    self._fill_next_state(pj_t, pa_t.to(torch.bool), ar)
            finally:
                leave(tok)
        k = -1
        while rec is None:
            k += 1
            tok = enter()
            try:
                if num_steps is None:                          # "until the end of every env's episode": one flag read per step
                    if bool(self._eoe.all()):
                        break
                    if int((self._head - self._tail).max()) >= self._cap:
                        self._cache_alloc(2 * self._cap)
                        c = self._c
                else:
                    if k == num_steps:
                        break
                    if k > 0 and k % max(1, reset_every) == 0:
                        if prev_j is not None:                 # the envs about to be finalised need their last element complete
                            self._fill_next_state(prev_j, prev_active, ar)
                            prev_j = None
                        roll_over()
                active = ~self._eoe
                dec = eng.decisions.clone()
                actor.act(acts, nact, decisions=dec, state=st_buf, choice=ch_buf)
                nact.mul_(active.to(torch.int32))
                state = st_buf.to(self.state_dtype)
                if prev_j is not None:                         # previous element: next_state = the state this step's gather produced,
                    still = prev_active & active               # or (episode over) the element's own state
                    pj = prev_j
                    cur = c["state"][ar, pj]
                    c["next_state"][ar, pj] = torch.where(still[:, None], state, torch.where(prev_active[:, None], cur, c["next_state"][ar, pj]))
                a1 = active[:, None]
                j = self._head & (self._cap - 1)
                agent = dec[:, 1].to(torch.int64).clamp(min=0)
                c["tick"][ar, j] = torch.where(active, dec[:, 0], c["tick"][ar, j])
                c["agent"][ar, j] = torch.where(active, agent, c["agent"][ar, j])
                c["state"][ar, j] = torch.where(a1, state, c["state"][ar, j])
                c["action"][ar, j] = torch.where(active, ch_buf.to(torch.int64), c["action"][ar, j])
                c["env_action"][ar, j] = torch.where(a1, acts[:, 0], c["env_action"][ar, j])
                c["terminal"][ar, j] = torch.where(active, torch.zeros_like(active), c["terminal"][ar, j])
                prev = self._last[ar, agent]                 # this agent's previous element gets its next agent state
                hp = active & (prev >= 0)
                pz = prev.clamp(min=0) & (self._cap - 1)
                c["next_agent_state"][ar, pz] = torch.where(hp[:, None], state, c["next_agent_state"][ar, pz])
                c["terminal"][ar, pz] = torch.where(hp, torch.zeros_like(hp), c["terminal"][ar, pz])
                self._last[ar, agent] = torch.where(active, self._head, self._last[ar, agent])
                self._head += active.to(torch.int64)
                self._steps_env += active.to(torch.int64)
                eng.step(acts, nact, mask=active.to(torch.uint8))
                self._eoe.copy_(torch.where(active, eng.done.to(torch.bool), self._eoe))
                prev_j, prev_active = j, active
            finally:
                leave(tok)
            yield k
        tok = enter()
        try:
            if prev_j is not None:
                self._fill_next_state(prev_j, prev_active, ar)
            self._cur_state = torch.where((~self._eoe)[:, None], self.state().to(self.state_dtype), self._cur_state)
            self._finalize_and_emit(torch.ones(n, dtype=torch.bool, device=dev), out, kernel=fused_kernels)
            res = {k: ((v[0] if len(v) == 1 else torch.cat(v)) if v else
                       torch.zeros((0,) + tuple(c[k].shape[2:]) if k in c else (0,), dtype=(c[k].dtype if k in c else torch.int32), device=dev))
                   for k, v in out.items()}
            if not out["reward"]:
                res["reward"] = torch.zeros(0, dtype=torch.float32, device=dev)
            res["env_metric"] = eng.metrics.clone()
            self._after_per_step_call()
        finally:
            leave(tok)
        return res

    # ------------------------------------------------------------------ the device-resident collection loop
    def _cache_struct(self):
        """The cache as the C ABI's argument block (include/maro_amd.h: mrx_cim_sampler_cache); rebuilt whenever an array is re-allocated."""
        from .. import _lib
        c, eng = self._c, self.eng
        key = tuple(int(v.data_ptr()) for v in c.values()) + (int(self._hist.data_ptr()), self._cap)
        if getattr(self, "_cs_key", None) != key:
            m = _lib.MrxCimSamplerCache()
            m.n_envs, m.n_ports, m.state_dim, m.cap = eng.n_envs, eng.layout.n_ports, self.state_dim, self._cap
            m.state_f64, m.window, m.frames = int(self.state_dtype == torch.float64), self.time_window, int(self._hist.shape[1])
            m.fulfillment_factor, m.shortage_factor = float(self.ff), float(self.sf)
            m.d_decay, m.d_eoe, m.d_head, m.d_tail, m.d_last = (t.data_ptr() for t in (self._decay, self._eoe, self._head, self._tail, self._last))
            m.d_prev_j, m.d_prev_active, m.d_interactions = self._pj.data_ptr(), self._pa.data_ptr(), self._steps_env.data_ptr()
            m.c_tick, m.c_agent, m.c_state, m.c_action = (c[k].data_ptr() for k in ("tick", "agent", "state", "action"))
            m.c_env_action, m.c_terminal, m.c_next_state, m.c_next_agent_state = (c[k].data_ptr() for k in ("env_action", "terminal", "next_state", "next_agent_state"))
            m.d_port_history = self._hist.data_ptr()
            self._cs, self._cs_key = m, key
        return self._cs

    def _sample_fused_device(self, actor, num_steps: int, seeds, reset_every: int, state_dtype: torch.dtype):
        """``sample_fused_steps`` with everything between two roll-over points on the device and nothing on the host:

        * a SEGMENT of interactions (up to `reset_every`) is one C call, ``actor.collect_steps`` = mrx_cim_collect_steps: per
          interaction the two policy launches — the transition-cache update rides in them — and the step launch;
        * the end of the call is mrx_cim_sampler_finalize (emission counts, their prefix over the envs, a 32-byte read-back) and
          mrx_cim_sampler_emit_all (block copies of the three state arrays, rewards from LDS-staged history rows); the generator
          yields between the two so that ``sample_fused_groups`` enqueues every group's finalisation before it waits for any;
        * ``_append_cache_element(None)`` and the last element's next state are applied where they become visible (emission, the
          next interaction) instead of eagerly for every env at the end of every call.

        Per call the host waits for the device once (the read-back), plus once per roll-over point that falls inside the call; the
        mid-call roll-over itself (rare: an env's episode ended) runs the tensor-op specification of the emission for those envs."""
        from .. import _lib
        from .engine import SEED_REDRAW
        eng, L = self.eng, self.eng._L
        n, dev = eng.n_envs, eng.decisions.device
        bound = getattr(eng, "_bound_stream", None)

        def enter():
            if bound is None:
                return None
            prev = torch.cuda.current_stream(bound.device)
            torch.cuda.set_stream(bound)
            return prev

        def leave(prev) -> None:
            if prev is not None:
                torch.cuda.set_stream(prev)

        def reset_envs(mask: torch.Tensor) -> None:   # AbsEnvSampler._reset for the envs in `mask` (a sync point: rare)
            cmd = seeds(self._ep_env.clone()).to(torch.int64) if seeds is not None else torch.full((n,), SEED_REDRAW, dtype=torch.int64)
            self._episodes += 1
            self._ep_env += mask.cpu().to(torch.int64)
            eng.reset(cmd, mask.to(torch.uint8))
            self._hist[mask] = 0
            self._tail.copy_(torch.where(mask, self._head, self._tail))
            self._last[mask] = -1
            self._pa[mask] = 0                           # no element waits for a next state
            eng.step(mask=mask.to(torch.uint8))          # _step(None): the first decision event
            self._eoe.copy_(torch.where(mask, eng.done.to(torch.bool), self._eoe))

        def roll_over(out) -> None:
            """Envs at the end of their episode: finalised, emitted (bound = final tick - delay), reset."""
            ended = self._eoe.clone()
            c = self._c
            rows = torch.nonzero(ended & self._pa.to(torch.bool)).view(-1)
            if rows.numel():   # their last element's next state is its own state (the binning launch does this at the next interaction)
                pj = self._pj[rows]
                c["next_state"][rows, pj] = c["state"][rows, pj]
                self._pa[rows] = 0
            self._finalize_and_emit(ended, out, kernel=True)
            reset_envs(ended)

        tok = enter()
        try:
            if not hasattr(self, "_c") or self.state_dtype != state_dtype:
                self._sample_init(state_dtype)
            if self._max_cached + num_steps + 1 > self._cap:      # host-side bound: no read-back
                self._cache_alloc(max(self._max_cached + num_steps + 1, 2 * self._cap))
            out = {k: [] for k in ("state", "action", "env_action", "reward", "next_state", "next_agent_state", "terminal", "env_id", "tick", "agent")}
            if not hasattr(self, "_acts"):
                self._acts = torch.zeros((n, eng.max_actions, 4), dtype=torch.int32, device=dev)
                self._nact = torch.zeros(n, dtype=torch.int32, device=dev)
                self._n_emit = torch.zeros(n, dtype=torch.int64, device=dev)
                self._off = torch.zeros(n, dtype=torch.int64, device=dev)
                self._info = torch.zeros(4, dtype=torch.int64, device=dev)
                self._info_host = torch.zeros(4, dtype=torch.int64).pin_memory()
            if self._any_eoe and bool(self._eoe.any()):          # (the flag is the previous call's read-back: usually no wait here)
                roll_over(out)
        finally:
            leave(tok)
        k = 0
        while k < num_steps:
            if k > 0 and k % max(1, reset_every) == 0:
                tok = enter()
                try:
                    torch.logical_or(self._eoe, eng.done, out=self._eoe)
                    if bool(self._eoe.any()):
                        roll_over(out)
                finally:
                    leave(tok)
            seg = min(num_steps - k, max(1, reset_every) - k % max(1, reset_every))
            actor.collect_steps(self._cache_struct(), self._acts, self._nact, seg)
            self._max_cached += seg
            k += seg
            yield k
        tok = enter()
        try:
            cs = self._cache_struct()
            _lib.check(L.mrx_cim_sampler_finalize(eng._h, ctypes.byref(cs), eng.done.data_ptr(), self._n_emit.data_ptr(), self._off.data_ptr(),
                                                  self._info.data_ptr(), eng._stream()), "mrx_cim_sampler_finalize")
            self._info_host.copy_(self._info, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
        finally:
            leave(tok)
        yield "finalized"     # (sample_fused_groups: the other groups enqueue their finalisation before anyone waits)
        tok = enter()
        try:
            ev.synchronize()
            K, max_cached, n_over, n_pending_out = (int(x) for x in self._info_host[:4].tolist())
            self._max_cached, self._any_eoe = max_cached, n_over > 0
            if n_pending_out:
                # a running env's newest element goes out in THIS call (its tick is a whole reward window behind the pending decision)
                # while its next state is still to come from the next interaction's gather: complete it now, as the reference does
                # right after _step (env_sampler.py:494-497).  Rare (sparse topologies / short windows); one state evaluation.
                self._fill_next_state(self._pj, self._pa.to(torch.bool), torch.arange(n, device=dev))
                self._pa.zero_()
            self._dev_pending = True   # (running envs' newest elements wait for their next state: see _drain_device_pending)
            D = self.state_dim
            if K > 0:
                o = dict(state=torch.empty((K, D), dtype=self.state_dtype, device=dev), action=torch.empty(K, dtype=torch.int64, device=dev),
                         env_action=torch.empty((K, 4), dtype=torch.int32, device=dev), reward=torch.empty(K, dtype=torch.float32, device=dev),
                         next_state=torch.empty((K, D), dtype=self.state_dtype, device=dev),
                         next_agent_state=torch.empty((K, D), dtype=self.state_dtype, device=dev), terminal=torch.empty(K, dtype=torch.bool, device=dev),
                         env_id=torch.empty(K, dtype=torch.int32, device=dev), tick=torch.empty(K, dtype=torch.int32, device=dev),
                         agent=torch.empty(K, dtype=torch.int32, device=dev))
                _lib.check(L.mrx_cim_sampler_emit_all(eng._h, ctypes.byref(cs), self._n_emit.data_ptr(), self._off.data_ptr(), o["state"].data_ptr(),
                                                      o["action"].data_ptr(), o["env_action"].data_ptr(), o["reward"].data_ptr(), o["next_state"].data_ptr(),
                                                      o["next_agent_state"].data_ptr(), o["terminal"].data_ptr(), o["env_id"].data_ptr(), o["tick"].data_ptr(),
                                                      o["agent"].data_ptr(), eng._stream()), "mrx_cim_sampler_emit_all")
                for key, v in o.items():
                    out[key].append(v)
            c = self._c
            res = {k2: ((v[0] if len(v) == 1 else torch.cat(v)) if v else
                        torch.zeros((0,) + tuple(c[k2].shape[2:]) if k2 in c else (0,), dtype=(c[k2].dtype if k2 in c else torch.int32), device=dev))
                   for k2, v in out.items()}
            if not out["reward"]:
                res["reward"] = torch.zeros(0, dtype=torch.float32, device=dev)
            res["env_metric"] = eng.metrics.clone()
        finally:
            leave(tok)
        return res

    def _drain_device_pending(self) -> None:
        """The device-resident loop (``_sample_fused_device``) ends a call with every running env's newest element still waiting for
        its next state (`_pa`): the next interaction's gather completes it.  The per-step paths (``sample``, ``sample_fused`` with
        `num_steps=None`, MRX_SAMPLER_V2=0, other state dtypes) neither read `_pa` nor had `_cur_state` maintained meanwhile — so
        a call that takes one of them after a device-path call completes those elements and refreshes `_cur_state` first."""
        if not getattr(self, "_dev_pending", False):
            return
        self._dev_pending = False
        dev = self.eng.decisions.device
        self._fill_next_state(self._pj, self._pa.to(torch.bool), torch.arange(self.eng.n_envs, device=dev))
        self._pa.zero_()
        self._cur_state = torch.where((~self._eoe)[:, None], self.state().to(self.state_dtype), self._cur_state)

    def _after_per_step_call(self) -> None:
        """The device-resident loop keeps two facts on the HOST between calls (so that a call needs no read-back before it starts): a
        bound on any env's cached elements and whether an env may sit at the end of its episode.  A per-step call changes both."""
        self._max_cached = int((self._head - self._tail).max())
        self._any_eoe = True

    @property
    def interactions(self) -> torch.Tensor:
        """int64 scalar tensor: env-steps this sampler has performed through sample_fused."""
        return self._steps_env.sum()

    def _fill_next_state(self, pj: torch.Tensor, p_active: torch.Tensor, ar: torch.Tensor) -> None:
        """next_state of the element every env wrote in its last step: the state of its now pending decision, or — episode
        over — the element's own state (`_cur_state` is left unchanged by a final step in ``sample``)."""
        c = self._c
        new = self.state().to(self.state_dtype)
        alive = p_active & ~self._eoe
        own = c["state"][ar, pj]
        c["next_state"][ar, pj] = torch.where(alive[:, None], new, torch.where(p_active[:, None], own, c["next_state"][ar, pj]))


def _seed_fn(seeds, n: int):
    """The `seeds` argument of sample / sample_fused*: a callable `seeds(episode_index) -> seed commands`.  `episode_index` is a
    CPU int64 tensor [n] — every env's OWN episode count (the reference runs one sampler loop per env: episode k of an env gets
    seed k of that env, whatever the rest of the batch is doing).  It may return an int64 tensor [n] on any device, or — for
    callers written against the single-env reference, `lambda ep: base + ep` style — anything that broadcasts to [n] on the CPU
    (a Python int, a 0-d or 1-element tensor).  Only the entries of the envs being reset are used."""
    if seeds is None:
        return None

    def fn(ep_env: torch.Tensor) -> torch.Tensor:
        cmd = seeds(ep_env)
        if not isinstance(cmd, torch.Tensor):
            cmd = torch.as_tensor(cmd, dtype=torch.int64)
        cmd = cmd.to(torch.int64)
        if cmd.numel() == 1 and n != 1:
            cmd = cmd.reshape(()).cpu().expand(n).contiguous()
        if cmd.shape != (n,):
            raise ValueError(f"seeds(episode_index int64 [{n}] on the CPU) must return {n} seed commands, got shape {tuple(cmd.shape)}")
        return cmd
    return fn


def sample_fused_groups(samplers, actors, num_steps: int, seeds=None, reset_every: int = 1, state_dtype: torch.dtype = torch.float32,
                        streams=None) -> list:
    """``sample_fused`` for the samplers of several env groups AT ONCE: their step generators are advanced in turn, each under
    its group's stream, so group g + 1's kernels are enqueued while group g's are running (a plain loop over
    ``sample_fused`` calls would finish one group — its closing emission reads sizes back — before the next one starts).
    `seeds`: one callable per group (or None).  Returns the per-group result dicts."""
    import contextlib
    G = len(samplers)
    gens = [samplers[g].sample_fused_steps(actors[g], num_steps, seeds=None if seeds is None else seeds[g], reset_every=reset_every,
                                           state_dtype=state_dtype) for g in range(G)]
    results, live = [None] * G, list(range(G))
    while live:
        for g in list(live):
            st = None if streams is None else streams[g]
            with (torch.cuda.stream(st) if st is not None else contextlib.nullcontext()):
                try:
                    next(gens[g])
                except StopIteration as stop:
                    results[g] = stop.value
                    live.remove(g)
    return results
