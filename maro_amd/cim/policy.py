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
        per = _lib.check(self._L.mrx_cim_dqn_net_floats(ctypes.byref(m)), "mrx_cim_dqn_net_floats")
        host = np.zeros((len(chains), per), dtype=np.float32)
        for p, chain in enumerate(chains):
            assert [w.shape for w, _ in chain] == [w.shape for w, _ in chains[0]], "all ports share one architecture"
            ws = [np.ascontiguousarray(w, dtype=np.float32) for w, _ in chain]
            bs = [np.ascontiguousarray(b, dtype=np.float32) for _, b in chain]
            wp = (ctypes.c_void_p * len(ws))(*[w.ctypes.data for w in ws])
            bp = (ctypes.c_void_p * len(bs))(*[b.ctypes.data for b in bs])
            _lib.check(self._L.mrx_cim_dqn_pack_net(ctypes.byref(m), wp, bp, host[p].ctypes.data), "mrx_cim_dqn_pack_net")
        self.weights = torch.from_numpy(host).to(engine.device)
        m.d_weights = self.weights.data_ptr()
        self._m = m
        self.scratch = torch.zeros(_lib.check(self._L.mrx_cim_dqn_scratch_bytes(engine._h), "mrx_cim_dqn_scratch_bytes"),
                                   dtype=torch.uint8, device=engine.device)

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
