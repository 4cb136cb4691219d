"""Device-side policy pieces of the CIM RL example (maro_amd/cim/policy.py): the action translation against a plain Python
restatement of examples/cim/rl/env_sampler.py:33-64, and the BatchNorm-folded dense chain against q-values produced by the
reference's own MyQNet (tests/golden/dqn_myqnet_small.npz, oracle/gen_golden_dqn.py)."""
import os
from collections import OrderedDict

import numpy as np
import torch

from maro_amd.cim.policy import ACTION_SPACE, PerPortDuelingQNet, dueling_chain, fold_fully_connected, translate_actions

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "dqn_myqnet_small.npz")


def ref_translate(model_action, load, discharge, vsl_space, early_discharge):
    percent = abs(ACTION_SPACE[model_action])
    zero_action_idx = len(ACTION_SPACE) / 2
    if model_action < zero_action_idx:
        return min(round(percent * load), vsl_space), 0
    if model_action > zero_action_idx:
        plan = percent * (discharge + early_discharge) - early_discharge
        return (round(plan) if plan > 0 else round(percent * discharge)), 1
    return 0, None


def test_translate_actions_matches_env_sampler():
    rng = np.random.RandomState(0)
    n = 5000
    dec = np.zeros((n, 8), np.int32)
    dec[:, 1], dec[:, 2] = rng.randint(0, 22, n), rng.randint(0, 46, n)
    dec[:, 3], dec[:, 4] = rng.randint(0, 5000, n), rng.randint(0, 5000, n)
    space, early = rng.randint(0, 4000, n), rng.randint(0, 300, n) * (rng.rand(n) < 0.3)
    ma = rng.randint(0, 21, n)
    got = translate_actions(torch.from_numpy(ma), torch.from_numpy(dec), torch.from_numpy(space), torch.from_numpy(early)).numpy()
    for i in range(n):
        q, ty = ref_translate(int(ma[i]), int(dec[i, 3]), int(dec[i, 4]), int(space[i]), int(early[i]))
        assert got[i, 0].tolist() == [dec[i, 2], dec[i, 1], q, ty], (i, ma[i], got[i], q, ty)


def fc_like_reference(dims, top=None, head_act=False):
    """A torch module with the structure (and state_dict keys) of maro.rl's FullyConnected (fc_block.py:72-133)."""
    def layer(i, o, act):
        mods = [("batch_norm", torch.nn.BatchNorm1d(i)), ("linear", torch.nn.Linear(i, o))]
        if act:
            mods.append(("activation", torch.nn.LeakyReLU()))
        return torch.nn.Sequential(OrderedDict(mods))

    layers = [layer(i, o, True) for i, o in zip(dims, dims[1:])]
    if top is not None:
        layers.append(layer(dims[-1], top, head_act))
    m = torch.nn.Module()
    m._net = torch.nn.Sequential(*layers)
    return m


def load_golden_net():
    g = np.load(GOLDEN)
    sd, A, hh = int(g["state_dim"]), int(g["action_num"]), int(g["head_hidden"])
    hidden = [int(h) for h in g["hidden"]]
    net = torch.nn.Module()
    net._fc = fc_like_reference([sd] + hidden)
    net._q = fc_like_reference([hidden[-1], hh], top=A)
    net._v = fc_like_reference([hidden[-1], hh], top=1)
    net.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd:")})
    net.eval()
    return g, net, A


def test_folded_chain_reproduces_reference_myqnet():
    g, net, A = load_golden_net()
    chain = dueling_chain(fold_fully_connected(net._fc), fold_fully_connected(net._q), fold_fully_connected(net._v))
    assert [w.shape for w, _ in chain] == [(45, 40), (40, 24), (24, 12), (12, 40), (40, A + 1)]
    x = g["states"].astype(np.float64)
    for i, (w, b) in enumerate(chain):   # numpy float64 evaluation of the folded chain
        x = x @ w.astype(np.float64) + b
        if i + 1 < len(chain):
            x = np.where(x > 0, x, 0.01 * x)
    q = x[:, :A] - x[:, :A].mean(axis=1, keepdims=True) + x[:, A:]
    # the reference evaluates BatchNorm and Linear separately in float32; folding reorders the arithmetic
    np.testing.assert_allclose(q, g["q"], rtol=2e-4, atol=2e-4)
    assert (q.argmax(axis=1) == g["greedy"]).all()
    # the torch float32 restatement used as the GPU tests' reference agrees as well
    tq = PerPortDuelingQNet([chain, chain], A)(torch.from_numpy(g["states"]), torch.zeros(64, dtype=torch.int64)).numpy()
    np.testing.assert_allclose(tq, g["q"], rtol=2e-4, atol=2e-4)


def test_per_port_qnet_selects_each_ports_network():
    from maro_amd.cim.policy import random_chains
    chains = random_chains(5, 19, 21, hidden=(32, 16), head_hidden=8, seed=1)
    net = PerPortDuelingQNet(chains, 21)
    x = torch.randn(40, 19)
    port = torch.arange(40) % 5
    q = net(x, port)
    assert q.shape == (40, 21)
    for i in (0, 7, 39):   # one env at a time through its own port's weights
        h = x[i:i + 1].numpy().astype(np.float64)
        for k, (w, b) in enumerate(chains[int(port[i])]):
            h = h @ w + b
            if k + 1 < len(chains[0]):
                h = np.where(h > 0, h, 0.01 * h)
        ref = h[:, :21] - h[:, :21].mean(axis=1, keepdims=True) + h[:, 21:]
        np.testing.assert_allclose(q[i].numpy(), ref[0], rtol=1e-4, atol=1e-4)


def test_chain_from_state_dict_equals_module_folding():
    """The reference ships state_dicts, not modules (policy_state = {name: policy.get_state()}, batch_env_sampler.py:150-176):
    folding straight from the dict's keys equals folding the loaded module, and the policy_state wrapper maps names to ports."""
    from maro_amd.cim.policy import chain_from_state_dict, chains_from_policy_state
    g, net, A = load_golden_net()
    want = dueling_chain(fold_fully_connected(net._fc), fold_fully_connected(net._q), fold_fully_connected(net._v))
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd:")}
    got = chain_from_state_dict(sd)
    assert len(got) == len(want)
    for (w, b), (w2, b2) in zip(got, want):
        assert np.array_equal(w, w2) and np.array_equal(b, b2)
    state = {"dqn_3.policy": {"net": {"network": sd, "optim": {}}, "policy": {"warmup": 0, "call_count": 0}}, "dqn_0.policy": {"net": {"network": sd}},
             7: sd}
    by_port = chains_from_policy_state(state)
    assert sorted(by_port) == [0, 3, 7] and all(np.array_equal(by_port[p][0][0], want[0][0]) for p in by_port)


def unpack_index(chains_like):
    """Invert mrx_cim_dqn_pack_net for tests: pack a chain whose every weight / bias holds its own serial number."""
    from maro_amd.cim.policy import pack_policy
    k = 1
    probe = []
    for w, b in chains_like:
        nw, nb = w.size, b.size
        probe.append((np.arange(k, k + nw, dtype=np.float32).reshape(w.shape), np.arange(k + nw, k + nw + nb, dtype=np.float32)))
        k += nw + nb
    assert k < 2 ** 24          # serial numbers stay exact in float32
    blob = pack_policy([probe], n_actions=chains_like[-1][0].shape[1] - 1)[0].numpy()
    pos = np.zeros(k, np.int64)
    nz = np.nonzero(blob)[0]
    pos[blob[nz].astype(np.int64)] = nz
    return pos[1:]


def unpack_policy(blob_row, chains_like, pos):
    flat = np.asarray(blob_row)[pos]
    out, k = [], 0
    for w, b in chains_like:
        out.append((flat[k:k + w.size].reshape(w.shape).copy(), flat[k + w.size:k + w.size + b.size].copy()))
        k += w.size + b.size
    return out


def test_pack_policy_is_a_permutation_of_the_parameters():
    from maro_amd.cim.policy import pack_policy, random_chains
    chains = random_chains(3, 45, 21, hidden=(40, 24, 12), head_hidden=20, seed=4)
    blob = pack_policy(chains)
    pos = unpack_index(chains[0])
    for p, chain in enumerate(chains):
        back = unpack_policy(blob[p].numpy(), chain, pos)
        assert all(np.array_equal(w, w2) and np.array_equal(b, b2) for (w, b), (w2, b2) in zip(back, chain))
