"""Batched device-side policy pieces of the CIM RL example: the per-port dueling DQN of
``examples/cim/rl/algorithms/dqn.py:13-84`` evaluated for a whole env batch, and the action translation of
``examples/cim/rl/env_sampler.py:33-64`` — so a rollout never leaves the GPU: state -> q-values -> env actions -> step.

* ``fold_fully_connected`` / ``dueling_chain`` turn the reference's network structure (``maro/rl/model/fc_block.py:72-133``:
  per layer BatchNorm1d -> Linear -> LeakyReLU, no activation on a ``head=True`` top layer) in eval mode into one chain of
  dense layers: BatchNorm folded into the linear map, the two dueling heads laid side by side (hidden layers
  concatenated, top layers block-diagonal) so the last width is ``n_actions + 1``.
* ``FusedPerPortDQN`` is the product path: ``mrx_cim_dqn_act`` (maro_amd/csrc/cim_dqn.h) — sampler state gather, the
  deciding port's network on exact-f32 MFMA, argmax and action translation in two launches.
* ``PerPortDuelingQNet`` + ``translate_actions`` are the plain PyTorch float32 restatement of the same computation
  (every network on every state, then a gather): the numerics reference of the tests, not used by the engine.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import _lib

ACTION_SPACE = [(i - 10) / 10 for i in range(21)]   # examples/cim/rl/config.py:20-24
PORT_ATTRIBUTES = ["empty", "full", "on_shipper", "on_consignee", "booking", "shortage", "fulfillment"]
VESSEL_ATTRIBUTES = ["empty", "full", "remaining_space"]

Chain = List[Tuple[np.ndarray, np.ndarray]]   # [(W float32 [in, out], b float32 [out])]


def fold_fully_connected(net: torch.nn.Module) -> Chain:
    """``FullyConnected._net`` (or the FullyConnected itself) in eval mode -> [(W [in, out], b [out])], float32.

    Each layer is ``Sequential(batch_norm?, linear, activation?)`` (fc_block.py:112-133); BatchNorm1d(x) =
    (x - mean) / sqrt(var + eps) * gamma + beta is folded into the following Linear in float64."""
    net = getattr(net, "_net", net)
    chain: Chain = []
    for layer in net:
        mods = dict(layer.named_children()) if not isinstance(layer, torch.nn.Linear) else {"linear": layer}
        lin = mods["linear"]
        w = lin.weight.detach().to(torch.float64).cpu().numpy().T.copy()           # [in, out]
        b = lin.bias.detach().to(torch.float64).cpu().numpy().copy() if lin.bias is not None else np.zeros(w.shape[1])
        bn = mods.get("batch_norm")
        if bn is not None:
            var, mean = bn.running_var.detach().double().cpu().numpy(), bn.running_mean.detach().double().cpu().numpy()
            gamma = bn.weight.detach().double().cpu().numpy() if bn.affine else np.ones_like(var)
            beta = bn.bias.detach().double().cpu().numpy() if bn.affine else np.zeros_like(var)
            s = gamma / np.sqrt(var + bn.eps)
            b = b + (beta - mean * s) @ w
            w = w * s[:, None]
        chain.append((w.astype(np.float32), b.astype(np.float32)))
    return chain


def dueling_chain(trunk: Chain, q_head: Chain, v_head: Chain) -> Chain:
    """MyQNet with dueling heads (dqn.py:33-52) as ONE dense chain: trunk, then the heads' hidden layers side by side and
    their top layers block-diagonal; the last layer's outputs are (advantages [n_actions], value [1])."""
    assert len(q_head) == len(v_head) and len(q_head) >= 1, "the two heads must have the same depth"
    out = list(trunk)
    for i, ((wq, bq), (wv, bv)) in enumerate(zip(q_head, v_head)):
        if i == 0:   # both read the trunk output
            w = np.concatenate([wq, wv], axis=1)
        else:        # block diagonal over the concatenated hidden vector
            w = np.zeros((wq.shape[0] + wv.shape[0], wq.shape[1] + wv.shape[1]), dtype=np.float32)
            w[:wq.shape[0], :wq.shape[1]] = wq
            w[wq.shape[0]:, wq.shape[1]:] = wv
        out.append((w.astype(np.float32), np.concatenate([bq, bv]).astype(np.float32)))
    return out


def chain_from_state_dict(sd, eps: float = 1e-5) -> Chain:
    """A ``MyQNet`` state_dict (examples/cim/rl/algorithms/dqn.py:25-52 — what ``AbsNet.get_state()["network"]`` carries,
    maro/rl/model/abs_net.py:65-74) -> the folded dense chain, without building the module: keys are
    ``<_fc|_q|_v>._net.<i>.<batch_norm|linear>.<weight|bias|running_mean|running_var>`` (fc_block.py:112-133).  Eval-mode
    BatchNorm is folded into the following Linear in float64 exactly as ``fold_fully_connected`` does; with ``_q`` / ``_v``
    present the heads are laid side by side (``dueling_chain``)."""
    def arr(k):
        v = sd[k]
        return (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)).astype(np.float64)

    def block(prefix) -> Chain:
        idx = sorted({int(k[len(prefix) + 6:].split(".")[0]) for k in sd if k.startswith(prefix + "._net.")})
        out: Chain = []
        for i in idx:
            base = f"{prefix}._net.{i}."
            w = arr(base + "linear.weight").T.copy()
            b = arr(base + "linear.bias") if base + "linear.bias" in sd else np.zeros(w.shape[1])
            if base + "batch_norm.running_var" in sd:
                var, mean = arr(base + "batch_norm.running_var"), arr(base + "batch_norm.running_mean")
                gamma = arr(base + "batch_norm.weight") if base + "batch_norm.weight" in sd else np.ones_like(var)
                beta = arr(base + "batch_norm.bias") if base + "batch_norm.bias" in sd else np.zeros_like(var)
                sc = gamma / np.sqrt(var + eps)
                b = b + (beta - mean * sc) @ w
                w = w * sc[:, None]
            out.append((w.astype(np.float32), b.astype(np.float32)))
        return out
    trunk = block("_fc")
    if any(k.startswith("_q._net.") for k in sd):
        return dueling_chain(trunk, block("_q"), block("_v"))
    return trunk


def chains_from_policy_state(policy_state: dict) -> dict:
    """The reference's ``policy_state`` — ``{policy_name: policy.get_state()}``, the dict ``BatchEnvSampler.sample`` ships to
    every rollout worker with each call (maro/rl/rollout/batch_env_sampler.py:150-176, worker.py:56-67) and
    ``AbsAgentWrapper.set_policy_state`` applies (env_sampler.py:37-46) — -> ``{port index: folded chain}``.  A policy's state
    is ``{"net": {"network": state_dict, "optim": ...}, "policy": ...}`` (discrete_rl_policy.py:217-225); bare state_dicts and
    ``{"network": ...}`` dicts are accepted too.  The port index is the agent index in the policy's name
    (``f"{algorithm}_{agent}.policy"``, examples/cim/rl/rl_component_bundle.py:20), or the key itself when it is an int."""
    out = {}
    for name, st in policy_state.items():
        if isinstance(name, (int, np.integer)):
            port = int(name)
        else:
            digits = [t for t in str(name).replace(".", "_").split("_") if t.isdigit()]
            if not digits:
                raise KeyError(f"cannot tell which port policy {name!r} belongs to (expected '<algorithm>_<agent>.policy')")
            port = int(digits[-1])
        sd = st.get("net", st) if isinstance(st, dict) else st
        sd = sd.get("network", sd) if isinstance(sd, dict) and "network" in sd else sd
        out[port] = chain_from_state_dict(sd)
    return out


def random_chains(n_ports: int, state_dim: int, action_num: int = 21, hidden: Sequence[int] = (256, 128, 64, 32),
                  head_hidden: int = 128, seed: int = 0) -> List[Chain]:
    """Random-init per-port dueling networks of the example's architecture (there are no trained checkpoints offline)."""
    rng = np.random.default_rng(seed)

    def lin(i, o):
        return (rng.standard_normal((i, o)) * (2.0 / i) ** 0.5).astype(np.float32), (rng.standard_normal(o) * 0.05).astype(np.float32)

    dims = [state_dim] + list(hidden)
    nets = []
    for _ in range(n_ports):
        trunk = [lin(dims[i], dims[i + 1]) for i in range(len(hidden))]
        nets.append(dueling_chain(trunk, [lin(dims[-1], head_hidden), lin(head_hidden, action_num)],
                                  [lin(dims[-1], head_hidden), lin(head_hidden, 1)]))
    return nets


def pack_policy(chains: Sequence[Chain], n_actions: int = len(ACTION_SPACE), dueling: bool = True) -> torch.Tensor:
    """Host only (no GPU, no engine): folded chains -> the packed MFMA operand blob, float32 [len(chains), per] on the CPU
    (mrx_cim_dqn_pack_net; the layout depends on the layer widths alone).  What a LEARNER rank — which owns no rollout engine —
    hands to ``rollout.broadcast_policy``; ``FusedPerPortDQN.pack`` is the same call on an actor's own model struct."""
    L = _lib.load()
    m = _lib.MrxCimDqnModel()
    dims = [chains[0][0][0].shape[0]] + [w.shape[1] for w, _ in chains[0]]
    m.n_nets, m.n_layers, m.dueling, m.n_actions = len(chains), len(chains[0]), int(dueling), int(n_actions)
    for i, d in enumerate(dims):
        m.dims[i] = d
    return _pack(L, m, chains, [tuple(w.shape) for w, _ in chains[0]])


def _pack(L, m, chains: Sequence[Chain], shapes) -> torch.Tensor:
    per = _lib.check(L.mrx_cim_dqn_net_floats(ctypes.byref(m)), "mrx_cim_dqn_net_floats")
    host = np.zeros((len(chains), per), dtype=np.float32)
    for p, chain in enumerate(chains):
        assert [tuple(w.shape) for w, _ in chain] == list(shapes), "all ports share one architecture"
        ws = [np.ascontiguousarray(w, dtype=np.float32) for w, _ in chain]
        bs = [np.ascontiguousarray(b, dtype=np.float32) for _, b in chain]
        wp = (ctypes.c_void_p * len(ws))(*[w.ctypes.data for w in ws])
        bp = (ctypes.c_void_p * len(bs))(*[b.ctypes.data for b in bs])
        _lib.check(L.mrx_cim_dqn_pack_net(ctypes.byref(m), wp, bp, host[p].ctypes.data), "mrx_cim_dqn_pack_net")
    return torch.from_numpy(host)


class FusedPerPortDQN:
    """``mrx_cim_dqn_act`` for one ``CimBatchEngine``: every deciding env's state -> its port's network -> greedy
    (or counter-based epsilon-greedy) action -> env action, written straight into the tensors the next ``step`` reads."""

    def __init__(self, engine, chains: Sequence[Chain], look_back: int = 7, port_attributes: Sequence[str] = PORT_ATTRIBUTES,
                 vessel_attributes: Sequence[str] = VESSEL_ATTRIBUTES, action_space: Sequence[float] = ACTION_SPACE,
                 dueling: bool = True, negative_slope: float = 0.01, epsilon: float = 0.0):
        self.eng, self._L = engine, _lib.load()
        m = _lib.MrxCimDqnModel()
        dims = [chains[0][0][0].shape[0]] + [w.shape[1] for w, _ in chains[0]]
        m.n_nets, m.n_layers = len(chains), len(chains[0])
        for i, d in enumerate(dims):
            m.dims[i] = d
        m.dueling, m.n_actions = int(dueling), len(action_space)
        m.negative_slope, m.epsilon, m.look_back = negative_slope, epsilon, look_back
        pa, va = engine.attr_ids("ports", port_attributes), engine.attr_ids("vessels", vessel_attributes)
        m.n_port_attrs, m.n_vessel_attrs = len(pa), len(va)
        for i, a in enumerate(pa):
            m.port_attrs[i] = a
        for i, a in enumerate(va):
            m.vessel_attrs[i] = a
        for i, a in enumerate(action_space):
            m.action_space[i] = float(a)
        self.state_dim, self.n_actions = dims[0], len(action_space)
        self._m, self._shapes = m, [tuple(w.shape) for w, _ in chains[0]]
        self._per = _lib.check(self._L.mrx_cim_dqn_net_floats(ctypes.byref(m)), "mrx_cim_dqn_net_floats")
        # the packed blob lives in ONE device tensor for the actor's lifetime (m.d_weights points into it): a learner refreshes
        # the policy by overwriting it in place (set_policy_state), never by re-allocating
        self.weights = self.pack(chains).to(engine.device)
        m.d_weights = self.weights.data_ptr()
        self.scratch = torch.zeros(_lib.check(self._L.mrx_cim_dqn_scratch_bytes(engine._h), "mrx_cim_dqn_scratch_bytes"),
                                   dtype=torch.uint8, device=engine.device)

    def pack(self, chains: Sequence[Chain]) -> torch.Tensor:
        """Host side of a policy update: folded chains -> the MFMA operand layout (mrx_cim_dqn_pack_net), float32 [len(chains), per]
        on the CPU.  What a learner broadcasts (`rollout.broadcast_policy`) and `set_policy_state` copies in."""
        return _pack(self._L, self._m, chains, self._shapes)

    def set_policy_state(self, state, ports: Optional[Sequence[int]] = None) -> None:
        """Refresh the networks IN PLACE — the sampler side of the reference's ``set_policy_state``
        (maro/rl/rollout/env_sampler.py:37-46; shipped with every ``sample`` request, batch_env_sampler.py:150-176).

        `state`: a packed blob (float32 [n_nets, per], CPU or device — what `pack` / `broadcast_policy` hand over), a list of
        folded chains (one per port, or per entry of `ports`), or the reference's ``{policy_name: policy.get_state()}`` dict
        (only the ports it names are touched).  No re-allocation: `weights` keeps its address, so ``mrx_cim_dqn_model.d_weights``
        stays valid.  Stream-ordered: the copy is enqueued on the engine's stream (the bound side stream, if any), after the
        caller's current stream when `state` is a device tensor — every `act` issued after this call sees the new weights,
        every `act` issued before it the old ones."""
        if isinstance(state, dict):
            by_port = chains_from_policy_state(state)
            ports = sorted(by_port)
            state = [by_port[p] for p in ports]
        if not isinstance(state, torch.Tensor):
            state = self.pack(state)
        rows = list(range(self.weights.shape[0])) if ports is None else [int(p) for p in ports]
        assert state.dtype == torch.float32 and tuple(state.shape) == (len(rows), self._per), \
            f"packed policy state must be float32 [{len(rows)}, {self._per}], got {state.dtype} {tuple(state.shape)}"
        bound = getattr(self.eng, "_bound_stream", None)
        dev = self.weights.device
        if state.is_cuda and bound is not None:
            bound.wait_stream(torch.cuda.current_stream(dev))
        import contextlib
        with (torch.cuda.stream(bound) if bound is not None else contextlib.nullcontext()):
            if ports is None:
                self.weights.copy_(state, non_blocking=True)
            else:
                self.weights[torch.as_tensor(rows, dtype=torch.int64, device=dev)] = state.to(dev, non_blocking=True)
            if not state.is_cuda and dev.type == "cuda":
                # a pageable host source may be reused by the caller as soon as this returns
                (bound or torch.cuda.current_stream(dev)).synchronize()

    def act(self, actions: torch.Tensor, n_actions: torch.Tensor, decisions: Optional[torch.Tensor] = None,
            q: Optional[torch.Tensor] = None, state: Optional[torch.Tensor] = None, choice: Optional[torch.Tensor] = None,
            counter: Optional[torch.Tensor] = None) -> None:
        """actions int32 [n, A, 4] / n_actions int32 [n] <- the policy's answer to `decisions` (default: engine.decisions).
        Optional outputs: q float32 [n, n_actions], state float32 [n, state_dim], choice int32 [n] (rows of deciding envs);
        counter int64 [1]: the number of answered decisions is added to it."""
        d = self.eng.decisions if decisions is None else decisions
        p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        _lib.check(self._L.mrx_cim_dqn_act(self.eng._h, ctypes.byref(self._m), d.data_ptr(), self.scratch.data_ptr(),
                                           actions.data_ptr(), n_actions.data_ptr(), p(q), p(state), p(choice), p(counter),
                                           self.eng._stream()), "mrx_cim_dqn_act")


    def collect_steps(self, cache, actions: torch.Tensor, n_actions: torch.Tensor, n_steps: int) -> None:
        """`n_steps` interactions of every env enqueued by ONE C call (mrx_cim_collect_steps): per interaction the two policy launches —
        with the batched EnvSampler's transition-cache update folded in (`cache`: a ``_lib.MrxCimSamplerCache`` over the sampler's
        device arrays) — and mrx_cim_step.  No host work between the interactions; results equal act -> record -> step."""
        e = self.eng
        _lib.check(self._L.mrx_cim_collect_steps(e._h, ctypes.byref(self._m), self.scratch.data_ptr(), ctypes.byref(cache), actions.data_ptr(),
                                                 n_actions.data_ptr(), e.decisions.data_ptr(), e.metrics.data_ptr(), e.done.data_ptr(), int(n_steps),
                                                 e._stream()), "mrx_cim_collect_steps")


class PerPortDuelingQNet(torch.nn.Module):
    """Plain PyTorch float32 restatement (test reference): q[n, A] of the network that belongs to each env's deciding port,
    from the same folded chains ``FusedPerPortDQN`` packs.  Every network is evaluated on every state, then gathered."""

    def __init__(self, chains: Sequence[Chain], n_actions: int, dueling: bool = True, negative_slope: float = 0.01):
        super().__init__()
        self.w = torch.nn.ParameterList([torch.nn.Parameter(torch.from_numpy(np.stack([c[i][0] for c in chains])), requires_grad=False)
                                         for i in range(len(chains[0]))])
        self.b = torch.nn.ParameterList([torch.nn.Parameter(torch.from_numpy(np.stack([c[i][1] for c in chains]))[:, None, :], requires_grad=False)
                                         for i in range(len(chains[0]))])
        self.n_actions, self.dueling, self.slope = n_actions, dueling, negative_slope

    @torch.no_grad()
    def forward(self, states: torch.Tensor, port: torch.Tensor, raw: bool = False) -> torch.Tensor:
        """q float [n, A] (module dtype: float32, or float64 after .double()); raw=True: the last layer's outputs
        [n, A + 1] before the dueling combination."""
        x = states.to(self.w[0].dtype).unsqueeze(0).expand(self.w[0].shape[0], -1, -1)
        for i, (w, b) in enumerate(zip(self.w, self.b)):
            x = torch.baddbmm(b, x, w)
            if i + 1 < len(self.w):
                x = torch.nn.functional.leaky_relu(x, self.slope)
        if self.dueling and not raw:
            adv, v = x[:, :, :self.n_actions], x[:, :, self.n_actions:]
            x = adv - adv.mean(dim=2, keepdim=True) + v                                          # dqn.py:48-52
        idx = port.to(torch.int64).view(1, -1, 1).expand(1, -1, x.shape[2])
        return x.gather(0, idx)[0]


def translate_actions(model_action: torch.Tensor, decisions: torch.Tensor, vessel_remaining_space: torch.Tensor,
                      vessel_early_discharge: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """env_sampler.py:33-64 for a batch: model_action int [n] (index into ACTION_SPACE), decisions int32 [n, 8] ->
    engine actions int32 [n, 1, 4] = (vessel_idx, port_idx, quantity, 0=LOAD | 1=DISCHARGE).  Index 10 (value 0.0) is a LOAD
    of 0 containers, as in the reference (zero_action_idx = 10.5).  Arithmetic in float64 with round-half-to-even, like
    Python's round()."""
    n = decisions.shape[0]
    load, discharge = decisions[:, 3].to(torch.float64), decisions[:, 4].to(torch.float64)
    # |ACTION_SPACE[a]| from a host-computed table: a device-side `/ 10.0` is evaluated as `* 0.1`, which is not the same double
    percent = torch.tensor([abs(a) for a in ACTION_SPACE], dtype=torch.float64, device=decisions.device)[model_action.to(torch.int64)]
    is_load = model_action < 10.5
    q_load = torch.minimum(torch.round(percent * load), vessel_remaining_space.to(torch.float64))
    early = vessel_early_discharge.to(torch.float64)
    plan = percent * (discharge + early) - early
    q_dis = torch.where(plan > 0, torch.round(plan), torch.round(percent * discharge))
    qty = torch.where(is_load, q_load, q_dis).to(torch.int32)
    if out is None:
        out = torch.zeros((n, 1, 4), dtype=torch.int32, device=decisions.device)
    out[:, 0, 0] = decisions[:, 2]
    out[:, 0, 1] = decisions[:, 1]
    out[:, 0, 2] = qty
    out[:, 0, 3] = (~is_load).to(torch.int32)
    return out
