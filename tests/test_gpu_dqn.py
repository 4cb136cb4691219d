"""mrx_cim_dqn_act (maro_amd/csrc/cim_dqn.h) on an MI355X: the fused sampler-state gather, per-port dueling DQN on f32 MFMA,
argmax and action translation — against the query-based CimBatchSampler state (bit-exact), plain PyTorch float32 /
float64 evaluations of the same networks (tolerance: |q - q64| <= 2e-5 x the magnitude of the last layer's outputs, and no
worse than 4x PyTorch's own float32 error) and translate_actions (bit-exact)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def run_case(topology, n, durations, steps, look_back, pa, va, hidden, head_hidden, seed, epsilon=0.0):
    from maro_amd.cim.engine import CimBatchEngine
    from maro_amd.cim.policy import ACTION_SPACE, FusedPerPortDQN, PerPortDuelingQNet, random_chains, translate_actions
    from maro_amd.cim.sampler import CimBatchSampler
    eng = CimBatchEngine(topology, n, durations=durations, seeds=torch.arange(n, dtype=torch.int64) * 7 + seed)
    smp = CimBatchSampler(eng, look_back=look_back, port_attributes=pa, vessel_attributes=va)
    chains = random_chains(eng.layout.n_ports, smp.state_dim, len(ACTION_SPACE), hidden=hidden, head_hidden=head_hidden, seed=seed)
    fused = FusedPerPortDQN(eng, chains, look_back=look_back, port_attributes=pa, vessel_attributes=va, epsilon=epsilon)
    ref = PerPortDuelingQNet(chains, len(ACTION_SPACE)).cuda()
    ref64 = PerPortDuelingQNet(chains, len(ACTION_SPACE)).double().cuda()
    A = len(ACTION_SPACE)
    actions = torch.zeros((n, 1, 4), dtype=torch.int32, device="cuda")
    n_actions = torch.zeros((n,), dtype=torch.int32, device="cuda")
    q = torch.zeros((n, A), dtype=torch.float32, device="cuda")
    st = torch.zeros((n, smp.state_dim), dtype=torch.float32, device="cuda")
    ch = torch.zeros((n,), dtype=torch.int32, device="cuda")
    eng.step()
    checked = 0
    for i in range(steps):
        d = eng.decisions.clone()
        valid = d[:, 7] == 1
        if not bool(valid.any()):
            break
        q.fill_(float("nan")), st.fill_(float("nan")), ch.fill_(-1), actions.fill_(-7)
        fused.act(actions, n_actions, q=q, state=st, choice=ch)
        torch.cuda.synchronize()
        assert torch.equal(n_actions, d[:, 7])
        want_state = smp.state(d).to(torch.float32)
        assert torch.equal(st[valid], want_state[valid]), f"state differs at step {i}"
        assert bool(torch.isnan(st[~valid]).all()) and bool((ch[~valid] == -1).all())   # untouched rows
        want_q = ref(want_state, d[:, 1].clamp(min=0))
        # float32 tolerance: the states are raw container counts (up to ~1e5) through random weights, so q = adv - mean + v
        # cancels large terms; the error bound is relative to the magnitude of the last layer's outputs, and the kernel must
        # be as close to a float64 evaluation as PyTorch's own float32 GEMMs are (x4 slack)
        q64 = ref64(want_state, d[:, 1].clamp(min=0))[valid]
        scale = float(ref64(want_state, d[:, 1].clamp(min=0), raw=True)[valid].abs().max())
        err_kernel, err_torch = float((q[valid].double() - q64).abs().max()), float((want_q[valid].double() - q64).abs().max())
        assert err_kernel <= 2e-5 * scale and err_kernel <= 4 * err_torch + 1e-6 * scale, (err_kernel, err_torch, scale)
        if epsilon == 0.0:
            assert torch.equal(ch[valid].to(torch.int64), q[valid].argmax(dim=1))
            top2 = want_q[valid].topk(2, dim=1).values            # same greedy action as the f32 reference unless a near tie
            clear = (top2[:, 0] - top2[:, 1]) > 1e-3
            assert torch.equal(ch[valid][clear].to(torch.int64), want_q[valid].argmax(dim=1)[clear])
        else:
            assert bool(((ch[valid] >= 0) & (ch[valid] < A)).all())
        vs = eng.query("vessels", d[:, 6:7], d[:, 2:3], ["remaining_space", "early_discharge"]).view(n, 2)
        want_a = translate_actions(ch.to(torch.int64), d, vs[:, 0], vs[:, 1])
        assert torch.equal(actions[valid], want_a[valid]), f"actions differ at step {i}"
        checked += int(valid.sum())
        eng.step(actions, n_actions)
    torch.cuda.synchronize()
    assert int(eng.status.max()) == 0, "the translated actions must all be legal"
    return checked


def test_fused_dqn_22p_example_architecture():
    """global_trade.22p with the example's own shapes: state 171 -> 256 -> 128 -> 64 -> 32 -> (128 | 128) -> (21 | 1)."""
    from maro_amd.cim.policy import PORT_ATTRIBUTES, VESSEL_ATTRIBUTES
    assert run_case("global_trade.22p_l0.8", 700, 120, 60, 7, PORT_ATTRIBUTES, VESSEL_ATTRIBUTES, (256, 128, 64, 32), 128, 3) > 20000


def test_fused_dqn_small_widths_and_few_ports():
    """4 ports (hundreds of envs per port -> many tiles per network, ragged last tiles), narrow layers (16- and 32-column paths)."""
    assert run_case("toy.4p_ssdd_l0.0", 1000, 150, 80, 3, ["empty", "shortage"], ["remaining_space"], (24, 8), 8, 5) > 20000


def test_fused_dqn_other_widths_and_epsilon():
    run_case("toy.5p_ssddd_l0.2", 130, 100, 40, 5, ["empty", "full", "booking"], ["empty", "full"], (192, 64, 17), 96, 9)
    run_case("toy.4p_ssdd_l0.0", 64, 100, 30, 4, ["transfer_cost", "empty"], ["early_discharge"], (48,), 20, 11, epsilon=0.5)


def test_set_policy_state_refreshes_the_networks_in_place():
    """A learner's update reaches a live actor without re-allocation (the reference: AbsAgentWrapper.set_policy_state,
    maro/rl/rollout/env_sampler.py:37-46): after set_policy_state the fused q-values are those of the NEW chains (= a fresh
    actor built from them, bit for bit), `weights` keeps its address, partial updates touch only the named ports, and the
    update is stream-ordered on an engine bound to a side stream."""
    from maro_amd.cim.engine import CimBatchEngine
    from maro_amd.cim.policy import ACTION_SPACE, FusedPerPortDQN, random_chains
    from maro_amd.cim.sampler import CimBatchSampler
    n, A = 300, len(ACTION_SPACE)
    eng = CimBatchEngine("toy.5p_ssddd_l0.5", n, durations=80, seeds=torch.arange(n, dtype=torch.int64) + 9)
    sd = CimBatchSampler(eng).state_dim
    arch = dict(hidden=(64, 32), head_hidden=16)
    old, new = (random_chains(5, sd, A, seed=s, **arch) for s in (1, 2))
    actor, fresh_new = FusedPerPortDQN(eng, old), FusedPerPortDQN(eng, new)
    mixed = FusedPerPortDQN(eng, [new[p] if p in (1, 3) else old[p] for p in range(5)])
    bufs = lambda: (torch.zeros((n, 1, 4), dtype=torch.int32, device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda"),  # noqa: E731
                    torch.zeros((n, A), dtype=torch.float32, device="cuda"))
    eng.step()
    for _ in range(5):
        a, na, q = bufs()
        actor.act(a, na, q=q)
        eng.step(a, na)
    addr = actor.weights.data_ptr()
    valid = eng.decisions[:, 7] == 1
    assert int(valid.sum()) > 100

    def qs(ac):
        a, na, q = bufs()
        ac.act(a, na, q=q)
        torch.cuda.synchronize()
        return q[valid].clone()
    q_old = qs(actor)
    actor.set_policy_state([new[1], new[3]], ports=[1, 3])                      # partial: chains for two ports
    assert torch.equal(qs(actor), qs(mixed)) and actor.weights.data_ptr() == addr
    actor.set_policy_state(actor.pack(new).cuda())                              # a device blob (what broadcast_policy hands over)
    assert torch.equal(qs(actor), qs(fresh_new)) and not torch.equal(qs(actor), q_old)
    actor.set_policy_state(old)                                                 # a list of chains
    assert torch.equal(qs(actor), q_old) and actor.weights.data_ptr() == addr
    # bound to a side stream: act issued right after the update sees the new weights without any explicit synchronisation
    st = torch.cuda.Stream()
    eng.use_stream(st)
    blob = actor.pack(new).cuda() * 1.0       # produced on the current stream
    actor.set_policy_state(blob)
    a, na, q = bufs()
    actor.act(a, na, q=q)
    st.synchronize()
    eng.use_stream(None)
    assert torch.equal(q[valid], qs(fresh_new))


def test_rccl_is_loaded_once_on_one_gpu():
    """`init_process_group("nccl", world_size=1)` in a child process (own timeout): RCCL loads, a communicator is created, and the
    device-tensor branches of broadcast_policy / gather_to_learner's size exchange run on it — so the first 8-GPU run is not the
    first time this build meets RCCL."""
    import os
    import subprocess
    import sys
    code = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
from maro_amd.cim.engine import CimBatchEngine
from maro_amd.cim.policy import ACTION_SPACE, FusedPerPortDQN, random_chains
from maro_amd.cim.rollout import broadcast_policy
from maro_amd.cim.sampler import CimBatchSampler
eng = CimBatchEngine("toy.4p_ssdd_l0.0", 64, durations=40, seeds=torch.arange(64))
sd = CimBatchSampler(eng).state_dim
old, new = (random_chains(4, sd, len(ACTION_SPACE), seed=s, hidden=(32, 16), head_hidden=8) for s in (1, 2))
actor = FusedPerPortDQN(eng, old)
want = actor.pack(new).cuda()
x = want.clone()
dist.broadcast(x, src=0)                       # RCCL collective on a device tensor
t = torch.ones(4, device="cuda"); dist.all_reduce(t); sizes = [torch.zeros(2, dtype=torch.int64, device="cuda")]
dist.all_gather(sizes, torch.tensor([64, 3], dtype=torch.int64, device="cuda"))
got = broadcast_policy(want, [actor], src=0)    # world 1: the local, in-place update
torch.cuda.synchronize()
assert torch.equal(x, want) and torch.equal(actor.weights, want) and sizes[0].tolist() == [64, 3] and t.tolist() == [1.0] * 4
print("RCCL_OK", dist.get_backend())
dist.destroy_process_group()
"""
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code, repo, "29631"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0 and "RCCL_OK nccl" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
