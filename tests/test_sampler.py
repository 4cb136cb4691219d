"""Batched observation / reward shaping (maro_amd/cim/sampler.py) against vectors produced by the reference's own
snapshot_list with the shaping code of examples/cim/rl/env_sampler.py (oracle/gen_golden.py, case *_sampler).
State vectors are exact (float64 copies of integer state); the reward is a float32 of a 99-term float64 dot
product, compared with rtol 1e-6 (summation order of torch.matmul vs numpy.dot)."""
import numpy as np
import pytest
import torch

from maro_amd.cim.sampler import CimBatchSampler
from tests.golden_util import case_topology, load_case, segment_actions


def run_sampler_case(engine_factory, n_envs=3):
    z, meta = load_case("toy5p_l05_sampler")
    topo = case_topology(meta)
    dur = meta["kwargs"]["durations"]
    eng = engine_factory(topo, n_envs, durations=dur, max_actions=2, seeds=[topo.seed] * n_envs)
    smp = CimBatchSampler(eng)
    gd, gs, gr = z["seg0/decisions"], z["seg0/sampler_state"], z["seg0/sampler_reward"]
    acts = segment_actions(z, 0)
    assert smp.state_dim == gs.shape[1] == 171
    dec, met, done = eng.step()
    A = 2
    for i in range(len(gd)):
        d = dec.cpu().numpy()
        assert np.array_equal(d[0], gd[i]) and np.array_equal(d[-1], gd[i])
        st = smp.state().cpu().numpy()
        assert st.dtype == np.float64 and np.array_equal(st[0], gs[i]) and np.array_equal(st[n_envs - 1], gs[i]), i
        a = np.full((n_envs, A, 4), -1, np.int32)
        na = np.zeros(n_envs, np.int32)
        for j, x in enumerate(acts[i]):
            a[:, j] = x
        na[:] = len(acts[i])
        dec, met, done = eng.step(a, na)
    assert bool(done.cpu().numpy().all())
    # delayed rewards of all decisions, evaluated at the end of the episode like AbsEnvSampler does
    dev = eng.decisions.device
    for i0 in range(0, len(gd), n_envs):
        rows = list(range(i0, min(i0 + n_envs, len(gd))))
        rows = rows + [rows[-1]] * (n_envs - len(rows))
        tick = torch.tensor(gd[rows, 0], dtype=torch.int32, device=dev)
        port = torch.tensor(gd[rows, 1], dtype=torch.int32, device=dev)
        r = smp.reward(tick, port).cpu().numpy()
        assert r.dtype == np.float32
        np.testing.assert_allclose(r, gr[rows], rtol=1e-6, atol=1e-6)


def emu_factory(topology, n, **kw):
    from tests.emu.emu_engine import EmuEngine
    return EmuEngine(topology, n, **kw)


def test_sampler_on_emulator():
    run_sampler_case(emu_factory)


@pytest.mark.gpu
def test_sampler_on_gpu():
    from maro_amd.cim.engine import CimBatchEngine
    run_sampler_case(lambda topo, n, **kw: CimBatchEngine(topo, n, **kw), n_envs=5)


# ---- the batched sampling loop against the REAL maro.rl sampler (oracle/gen_golden_sampler.py)
def run_sample_case(engine_factory, case, n_envs=2, fused=False):
    """CimBatchSampler.sample(num_steps) called like the golden's CIMEnvSampler.sample: same seeds, the recorded model actions
    replayed as the policy; every emitted experience (tick, agent, state, action, reward, terminal, next_state,
    next_agent_state) of every env must equal the reference's, call by call."""
    import json
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"sampler_{case}.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    eng = engine_factory(meta["topology"], n_envs, durations=meta["durations"], max_actions=1, max_snapshots=16)
    smp = CimBatchSampler(eng)
    assert smp.state_dim == meta["state_dim"]
    inter = z["interactions"]
    k = [0]
    dev = eng.decisions.device

    def policy(states, dec):
        a = int(inter[k[0], 1])
        d = dec.cpu().numpy()
        live = d[:, 7] == 1
        assert (d[live, 1] == inter[k[0], 0]).all() and (d[live, 0] == inter[k[0], 6]).all(), k[0]   # same agent, same tick as the reference
        k[0] += 1
        return torch.full((n_envs,), a, dtype=torch.int64, device=dev)

    def seeds(ep):   # ep: int64 [n_envs], every env's own episode index
        return meta["seed"] + ep.to(torch.int64)

    class RecordedActor:
        """`actor.act` of CimBatchSampler.sample_fused (what FusedPerPortDQN does in one launch on the GPU), from the pieces the
        unfused loop uses: the sampler state, the recorded model action, the example's action translation."""

        def act(self, actions, n_actions, decisions=None, state=None, choice=None):
            from maro_amd.cim.policy import translate_actions
            st = smp.state(decisions)
            ma = policy(st, decisions)
            translate_actions(ma, decisions, st[:, -1].to(torch.float64), decisions[:, 5], out=actions)
            n_actions[:] = (decisions[:, 7] == 1).to(torch.int32)
            state[:] = st.to(torch.float32)
            choice[:] = ma.to(torch.int32)

    for c, num_steps in enumerate(meta["calls"]):
        if fused:
            res = smp.sample_fused(RecordedActor(), num_steps=num_steps, seeds=seeds, reset_every=1, state_dtype=torch.float64)
        else:
            res = smp.sample(policy, num_steps=num_steps, seeds=seeds, state_dtype=torch.float64)
        assert k[0] == int(z[f"call{c}/interactions"][1]), (c, k[0])
        n_exp = len(z[f"call{c}/tick"])
        env_id = res["env_id"].cpu().numpy()
        for e in range(n_envs):
            sel = np.flatnonzero(env_id == e)
            assert len(sel) == n_exp, (c, e, len(sel), n_exp)
            if n_exp == 0:
                continue
            g = {key: res[key].cpu().numpy()[sel] for key in ("tick", "agent", "state", "action", "reward", "terminal", "next_state", "next_agent_state")}
            assert np.array_equal(g["tick"], z[f"call{c}/tick"]) and np.array_equal(g["agent"], z[f"call{c}/agent"])
            assert np.array_equal(g["action"], z[f"call{c}/action"]) and np.array_equal(g["terminal"], z[f"call{c}/terminal"])
            assert np.array_equal(g["state"], z[f"call{c}/state"]) and np.array_equal(g["next_state"], z[f"call{c}/next_state"])
            assert np.array_equal(g["next_agent_state"], z[f"call{c}/next_agent_state"])
            np.testing.assert_allclose(g["reward"], z[f"call{c}/reward"], rtol=1e-6, atol=1e-6)   # float32 of a 99-term float64 dot product
        env_act = res["env_action"].cpu().numpy()
        assert bool(eng.done.cpu().numpy().all()) == bool(z[f"call{c}/end_of_episode"][0])
    assert k[0] == len(inter)


@pytest.mark.parametrize("case", ["toy5p_l05", "gt22p_l08", "toy5p_l05_rollover", "toy6p_l08", "toy4p_l00_rollover"])
def test_batched_sample_matches_the_reference_sampler_on_emulator(case):
    run_sample_case(emu_factory, case)


@pytest.mark.parametrize("case", ["toy5p_l05", "gt22p_l08", "toy5p_l05_rollover", "toy6p_l08", "toy4p_l00_rollover"])
def test_fused_sample_loop_matches_the_reference_sampler_on_emulator(case):
    """sample_fused (the sync-free loop bench.py --policy dqn --collect times), reset_every = 1: the same experiences."""
    run_sample_case(emu_factory, case, fused=True)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["toy5p_l05", "gt22p_l08", "toy5p_l05_rollover", "toy6p_l08", "toy4p_l00_rollover"])
def test_batched_sample_matches_the_reference_sampler_on_gpu(case):
    from maro_amd.cim.engine import CimBatchEngine
    run_sample_case(lambda topo, n, **kw: CimBatchEngine(topo, n, **kw), case, n_envs=7)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["toy5p_l05", "gt22p_l08", "toy5p_l05_rollover", "toy6p_l08", "toy4p_l00_rollover"])
def test_fused_sample_loop_matches_the_reference_sampler_on_gpu(case):
    """... and the fused loop on the HIP build: mrx_cim_sampler_record + mrx_cim_sampler_emit against the reference's experiences."""
    from maro_amd.cim.engine import CimBatchEngine
    run_sample_case(lambda topo, n, **kw: CimBatchEngine(topo, n, **kw), case, n_envs=7, fused=True)


@pytest.mark.parametrize("fused", [False, True])
def test_divergent_episodes_in_one_batch_equal_single_env_runs(fused):
    """Envs with different seeds reach their episode ends at different steps: each env's roll-over must use ITS OWN episode index
    for the next seed (the reference runs one sampler loop per env), and the partial-mask reset / emit paths must leave the other
    envs alone.  A 3-env batch against three 1-env samplers, with and without the fused loop."""
    topo, dur, n = "toy.5p_ssddd_l0.5", 60, 3
    base = [11, 500, 9001]

    def policy(states, dec):
        return ((dec[:, 0] + 3 * dec[:, 1]) % 21).to(torch.int64)

    class Actor:
        def __init__(self, smp):
            self.smp = smp

        def act(self, actions, n_actions, decisions=None, state=None, choice=None):
            from maro_amd.cim.policy import translate_actions
            st = self.smp.state(decisions)
            ma = policy(st, decisions)
            translate_actions(ma, decisions, st[:, -1].to(torch.float64), decisions[:, 5], out=actions)
            n_actions[:] = (decisions[:, 7] == 1).to(torch.int32)
            state[:] = st.to(torch.float32)
            choice[:] = ma.to(torch.int32)

    def run(env_bases):
        m = len(env_bases)
        eng = emu_factory(topo, m, durations=dur, max_actions=1, max_snapshots=16)
        smp = CimBatchSampler(eng, time_window=20)
        b = torch.tensor(env_bases, dtype=torch.int64)
        res = []
        for num_steps in (25, 40, 33):
            if fused:
                res.append(smp.sample_fused(Actor(smp), num_steps=num_steps, seeds=lambda ep: b + 7 * ep, reset_every=1, state_dtype=torch.float64))
            else:
                res.append(smp.sample(policy, num_steps=num_steps, seeds=lambda ep: b + 7 * ep, state_dtype=torch.float64))
        return res

    batch = run(base)
    assert any(len(r["tick"]) for r in batch)
    for e, s in enumerate(base):
        single = run([s])
        for rb, rs in zip(batch, single):
            sel = np.flatnonzero(rb["env_id"].numpy() == e)
            for key in ("tick", "agent", "state", "action", "env_action", "reward", "terminal", "next_state", "next_agent_state"):
                assert np.array_equal(rb[key].numpy()[sel], rs[key].numpy()), (e, key)


def _grouped_sampling_case(engine_factory, streams=None, dur=60):
    """sample_fused_groups (the groups' step generators advanced in turn — what bench.py --collect drives) against one
    sample_fused call per group: the same experiences, group by group, over three calls with episode roll-overs."""
    from maro_amd.cim.sampler import sample_fused_groups
    topo, sizes = "toy.5p_ssddd_l0.5", (3, 2, 4)

    class Actor:
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

    def make(on_streams=None):
        import contextlib
        smps = []
        for g, m in enumerate(sizes):
            st = None if on_streams is None else on_streams[g]
            with (torch.cuda.stream(st) if st is not None else contextlib.nullcontext()):
                eng = engine_factory(topo, m, durations=dur, max_actions=1, max_snapshots=16)
                if st is not None:
                    eng.use_stream(st)      # engine kernels and the sampler's tensor ops on the group's stream, as bench.py does
                smps.append(CimBatchSampler(eng, time_window=20))
        seeds = [(lambda ep, g=g, m=m: 100 * g + 7 * ep + torch.arange(m, dtype=torch.int64)) for g, m in enumerate(sizes)]
        return smps, [Actor(s) for s in smps], seeds

    a_s, a_act, a_seed = make(streams)
    b_s, b_act, b_seed = make()
    if streams is not None:
        torch.cuda.synchronize()
    total = 0
    for num_steps in (25, 40, 33):
        ra = sample_fused_groups(a_s, a_act, num_steps, seeds=a_seed, reset_every=4, state_dtype=torch.float64, streams=streams)
        rb = [s.sample_fused(act, num_steps=num_steps, seeds=sd, reset_every=4, state_dtype=torch.float64) for s, act, sd in zip(b_s, b_act, b_seed)]
        for x, y in zip(ra, rb):
            assert set(x) == set(y)
            for key in x:
                assert torch.equal(x[key], y[key]), key
            total += len(x["tick"])
    assert total > 100


def test_grouped_sampling_equals_one_call_per_group_on_emulator():
    _grouped_sampling_case(emu_factory)


@pytest.mark.gpu
def test_grouped_sampling_equals_one_call_per_group_on_gpu():
    from maro_amd.cim.engine import CimBatchEngine
    _grouped_sampling_case(lambda topo, m, **kw: CimBatchEngine(topo, m, **kw), streams=[torch.cuda.Stream() for _ in range(3)])


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("case,slots", [("toy5p_l05_rollover", 2), ("gt22p_l08", 8), ("toy4p_l00_rollover", 4)])
def test_transition_ring_wraps_and_grows(case, slots, fused, monkeypatch):
    """The per-env transition cache is a ring: started with a few slots it wraps around many times and doubles whenever an
    env's live elements no longer fit — the emitted experiences stay the reference's."""
    monkeypatch.setattr(CimBatchSampler, "INITIAL_CACHE_SLOTS", slots)
    grown = []
    orig = CimBatchSampler._cache_alloc

    def spy(self, cap):
        grown.append(cap)
        return orig(self, cap)
    monkeypatch.setattr(CimBatchSampler, "_cache_alloc", spy)
    run_sample_case(emu_factory, case, fused=fused)
    assert len(grown) >= 3 and grown[0] == slots, grown


@pytest.mark.gpu
def test_fused_collect_loop_with_the_fused_dqn_on_gpu():
    """bench.py --policy dqn --collect's loop (sample_fused + FusedPerPortDQN.act) against the unfused sample() driven by the SAME
    network (its policy callable reads the fused kernel's model action): identical experiences with reset_every = 1; with
    reset_every = 32 every env's own stream of experiences is a prefix-equal subsequence (only the alignment across the batch moves)."""
    from maro_amd.cim.engine import CimBatchEngine
    from maro_amd.cim.policy import ACTION_SPACE, FusedPerPortDQN, random_chains
    topo, n, dur = "global_trade.22p_l0.8", 96, 160
    seeds = lambda ep: 77 + 13 * ep + torch.arange(n, dtype=torch.int64)   # noqa: E731

    def make():
        eng = CimBatchEngine(topo, n, durations=dur, max_actions=1, max_snapshots=16)
        smp = CimBatchSampler(eng, time_window=30)
        chains = random_chains(eng.topo.n_ports, smp.state_dim, len(ACTION_SPACE), seed=3)
        return eng, smp, FusedPerPortDQN(eng, chains)

    def run(mode):
        eng, smp, q = make()
        a = torch.zeros((n, 1, 4), dtype=torch.int32, device=eng.device)
        na = torch.zeros(n, dtype=torch.int32, device=eng.device)
        ch = torch.zeros(n, dtype=torch.int32, device=eng.device)

        def policy(states, dec):
            q.act(a, na, decisions=dec, choice=ch)
            return ch.to(torch.int64)
        out = []
        for k in (50, 70, 64, 90):
            if mode == "unfused":
                out.append(smp.sample(policy, num_steps=k, seeds=seeds, state_dtype=torch.float32))
            else:
                out.append(smp.sample_fused(q, num_steps=k, seeds=seeds, reset_every=mode, state_dtype=torch.float32))
        return out

    ref, f1, f32 = run("unfused"), run(1), run(32)
    CimBatchSampler.INITIAL_CACHE_SLOTS, keep = 16, CimBatchSampler.INITIAL_CACHE_SLOTS
    try:
        small = run(1)            # the record kernel on a ring that wraps and is re-allocated between the calls
    finally:
        CimBatchSampler.INITIAL_CACHE_SLOTS = keep
    keys = ("env_id", "tick", "agent", "state", "action", "env_action", "reward", "terminal", "next_state", "next_agent_state")
    total = 0
    def same(a, b, key):    # rewards: the emit kernel sums the same float64 terms in another order than the tensor ops
        return torch.allclose(a, b, rtol=1e-6, atol=1e-6) if key == "reward" else torch.equal(a, b)
    for r, f, g in zip(ref, f1, small):
        for key in keys:
            assert same(r[key], f[key], key) and same(r[key], g[key], key), key
        total += len(r["tick"])
    assert total > 1000

    def per_env(res):
        cat = {k: torch.cat([r[k] for r in res]).cpu().numpy() for k in keys}
        return [{k: cat[k][cat["env_id"] == e] for k in keys} for e in range(n)]
    a_env, b_env = per_env(ref), per_env(f32)
    for e in range(n):
        m = min(len(a_env[e]["tick"]), len(b_env[e]["tick"]))
        assert m > 5
        for key in keys:
            if key == "reward":
                np.testing.assert_allclose(a_env[e][key][:m], b_env[e][key][:m], rtol=1e-6, atol=1e-6)
            else:
                assert np.array_equal(a_env[e][key][:m], b_env[e][key][:m]), (e, key)


def test_bench_collect_parity_helper_on_emulator():
    """tests/bench_parity.py::replay_collect_against_oracle — the check bench.py's config-5 leg reports as `parity` — on two
    emulator-backed groups: the loop's own elements drive the C oracle, states / translations / delayed rewards must agree; and a
    corrupted element is caught."""
    from tests.bench_parity import replay_collect_against_oracle
    from tests.test_distributed_gloo import _Actor
    topo, dur = "toy.5p_ssddd_l0.5", 600
    engs = [emu_factory(topo, n, durations=dur, max_actions=1, max_snapshots=16) for n in (3, 2)]
    smps = [CimBatchSampler(e) for e in engs]
    actors = [_Actor(s) for s in smps]
    seeds_of = [(lambda ep, g=g, n=e.n_envs: 500 + 13 * ep + torch.arange(n, dtype=torch.int64) + 10 * g) for g, e in enumerate(engs)]
    rep = replay_collect_against_oracle(smps, actors, seeds_of, [0, 3], topo, k=4, num_steps=140, reset_every=8)
    assert rep["ok"] and rep["envs_checked"] == 4 and rep["elements_checked"] == 560 and rep["rewards_checked"] > 40, rep

    class Skewed(_Actor):      # a wrong translation (one container too many on every 7th tick) must be reported
        def act(self, actions, n_actions, decisions=None, state=None, choice=None):
            super().act(actions, n_actions, decisions=decisions, state=state, choice=choice)
            actions[:, 0, 2] += ((decisions[:, 0] % 7 == 0) & (actions[:, 0, 2] > 0)).to(torch.int32) * -1
    bad = replay_collect_against_oracle(smps, [Skewed(s) for s in smps], seeds_of, [0, 3], topo, k=2, num_steps=60, reset_every=8)
    assert not bad["ok"] and bad["first_mismatch"] is not None


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_device_resident_collection_loop_equals_the_tensor_path(dtype):
    """sample_fused with a FusedPerPortDQN actor runs the device-resident loop (mrx_cim_collect_steps: cache update folded into the
    policy launches; mrx_cim_sampler_finalize / emit_all at the end of the call).  It must emit exactly what the per-step path
    (act -> mrx_cim_sampler_record -> step, tensor-op finalisation: pinned to the reference sampler's goldens above) emits — every
    field of every experience, call by call, through several episode roll-overs, cache growth and mid-call roll-over points."""
    import os
    from maro_amd.cim.engine import CimBatchEngine
    from maro_amd.cim.policy import ACTION_SPACE, FusedPerPortDQN, random_chains
    topo, n, dur, calls = "toy.5p_ssddd_l0.5", 61, 230, (25, 90, 7, 160, 33, 1, 120)
    res = {}
    for mode in ("1", "0"):
        os.environ["MRX_SAMPLER_V2"] = mode
        try:
            eng = CimBatchEngine(topo, n, durations=dur, max_actions=1, max_snapshots=16)
            smp = CimBatchSampler(eng, time_window=40)
            smp.INITIAL_CACHE_SLOTS = 32          # (forces the ring to grow between calls)
            actor = FusedPerPortDQN(eng, random_chains(5, smp.state_dim, len(ACTION_SPACE), hidden=(48, 16), head_hidden=8, seed=3), epsilon=0.3)
            seeds = lambda ep: 4000 + 17 * ep + torch.arange(n, dtype=torch.int64)   # noqa: E731
            res[mode] = [smp.sample_fused(actor, num_steps=k, seeds=seeds, reset_every=8, state_dtype=dtype) for k in calls]
            torch.cuda.synchronize()
            assert int(eng.status.abs().sum()) == 0
            res[mode + "steps"], res[mode + "episodes"] = int(smp.interactions.item()), int(smp._ep_env.max())
        finally:
            os.environ.pop("MRX_SAMPLER_V2", None)
    assert res["1steps"] == res["0steps"] > 0 and res["1episodes"] == res["0episodes"] >= 3      # (several roll-overs)
    total = 0
    for c, (a, b) in enumerate(zip(res["1"], res["0"])):
        assert set(a) == set(b)
        for key in b:
            assert a[key].shape == b[key].shape and a[key].dtype == b[key].dtype, (c, key, a[key].shape, b[key].shape)
            if key == "reward":
                torch.testing.assert_close(a[key], b[key], rtol=1e-6, atol=1e-6)
            else:
                assert torch.equal(a[key], b[key]), (c, key)
        total += int(b["tick"].shape[0])
    assert total > 20000


def _compare_calls(res_a, res_b):
    total = 0
    for c, (a, b) in enumerate(zip(res_a, res_b)):
        assert set(a) == set(b)
        for key in b:
            assert a[key].shape == b[key].shape and a[key].dtype == b[key].dtype, (c, key, a[key].shape, b[key].shape)
            if key == "reward":
                torch.testing.assert_close(a[key], b[key], rtol=1e-6, atol=1e-6)
            else:
                assert torch.equal(a[key], b[key]), (c, key)
        total += int(b["tick"].shape[0])
    return total


@pytest.mark.gpu
def test_per_step_paths_after_a_device_resident_call_complete_the_pending_elements():
    """the device-resident loop ends a call with every running env's newest element waiting for its next state; a
    later call on the SAME sampler through a per-step path (`num_steps=None`, `sample`) must complete it first.  One sampler
    alternates device-resident and per-step calls, the other takes the per-step path throughout: identical experiences, call by call."""
    import os
    from maro_amd.cim.engine import CimBatchEngine
    from maro_amd.cim.policy import ACTION_SPACE, FusedPerPortDQN, random_chains
    topo, n, dur, calls = "toy.5p_ssddd_l0.5", 37, 150, (25, None, 40, 13, None, 60)
    res = {}
    for mode in ("1", "0"):
        os.environ["MRX_SAMPLER_V2"] = mode
        try:
            eng = CimBatchEngine(topo, n, durations=dur, max_actions=1, max_snapshots=16)
            smp = CimBatchSampler(eng, time_window=25)
            actor = FusedPerPortDQN(eng, random_chains(5, smp.state_dim, len(ACTION_SPACE), hidden=(32, 16), head_hidden=8, seed=5), epsilon=0.3)
            seeds = lambda ep: 700 + 13 * ep + torch.arange(n, dtype=torch.int64)   # noqa: E731
            res[mode] = [smp.sample_fused(actor, num_steps=k, seeds=seeds, reset_every=8) for k in calls]
            torch.cuda.synchronize()
            assert int(eng.status.abs().sum()) == 0
        finally:
            os.environ.pop("MRX_SAMPLER_V2", None)
    assert _compare_calls(res["1"], res["0"]) > 2000


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_device_resident_loop_with_a_window_shorter_than_the_gaps_between_decisions(dtype):
    """with a short reward window a running env's newest element can be old enough to be emitted in the call that wrote
    it, before the next interaction has supplied its next state — the device path then completes it on the host before the
    emission (mrx_k_cim_sampler_scan's info[3]).  toy.4p_ssdd_l0.0: decisions several ticks apart; window 2."""
    import os
    from maro_amd.cim.engine import CimBatchEngine
    from maro_amd.cim.policy import ACTION_SPACE, FusedPerPortDQN, random_chains
    topo, n, dur, calls = "toy.4p_ssdd_l0.0", 29, 200, (5, 1, 17, 3, 40, 2, 9)
    res, late = {}, 0
    for mode in ("1", "0"):
        os.environ["MRX_SAMPLER_V2"] = mode
        try:
            eng = CimBatchEngine(topo, n, durations=dur, max_actions=1, max_snapshots=16)
            smp = CimBatchSampler(eng, time_window=2)
            actor = FusedPerPortDQN(eng, random_chains(4, smp.state_dim, len(ACTION_SPACE), hidden=(32, 16), head_hidden=8, seed=9), epsilon=0.3)
            seeds = lambda ep: 90 + 7 * ep + torch.arange(n, dtype=torch.int64)   # noqa: E731
            out = []
            for k in calls:
                out.append(smp.sample_fused(actor, num_steps=k, seeds=seeds, reset_every=4, state_dtype=dtype))
                if mode == "1":
                    late += int(smp._info_host[3])
            res[mode] = out
            torch.cuda.synchronize()
            assert int(eng.status.abs().sum()) == 0
        finally:
            os.environ.pop("MRX_SAMPLER_V2", None)
    assert late > 0, "the case this test is about did not occur"
    assert _compare_calls(res["1"], res["0"]) > 300


# ---- CimBatchSampler.eval against the REAL CIMEnvSampler.eval (oracle/gen_golden_sampler_eval.py)
def run_eval_case(engine_factory, case, n_envs=2):
    """eval(num_episodes) called like the golden's: same test-env seeds, the recorded exploiting model actions replayed as the
    policy; every episode's env_metric of every env must equal the reference's ``info["env_metric"]``."""
    import json
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"sampler_eval_{case}.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    eng = engine_factory(meta["topology"], n_envs, durations=meta["durations"], max_actions=1, max_snapshots=16)
    smp = CimBatchSampler(eng)
    inter, k, dev = z["interactions"], [0], eng.decisions.device

    def policy(states, dec):
        d = dec.cpu().numpy()
        live = d[:, 7] == 1
        if not live.any():                      # (interactions enqueued past the batch's last decision)
            return torch.zeros(n_envs, dtype=torch.int64, device=dev)
        assert live.all() and (d[:, 1] == inter[k[0], 0]).all() and (d[:, 0] == inter[k[0], 6]).all(), k[0]
        a = int(inter[k[0], 1])
        k[0] += 1
        return torch.full((n_envs,), a, dtype=torch.int64, device=dev)

    res = smp.eval(policy, num_episodes=meta["episodes"], seeds=lambda ep: meta["seed"] + ep.to(torch.int64), done_every=1)
    assert k[0] == len(inter) and len(res["info"]) == meta["episodes"]
    for ep, info in enumerate(res["info"]):
        m = info["env_metric"].cpu().numpy()
        assert m.shape == (n_envs, 3) and (m == z["env_metric"][ep][None, :]).all(), (ep, m, z["env_metric"][ep])
    # a sample() after the eval starts from fresh episodes (the engine's episodes are over)
    out = smp.sample(lambda s, d: torch.zeros(n_envs, dtype=torch.int64, device=dev), num_steps=3)
    assert out["env_metric"].shape == (n_envs, 3) and not bool(eng.done.any())


@pytest.mark.parametrize("case", ["toy5p_l05", "gt22p_l08"])
def test_batched_eval_matches_the_reference_sampler_on_emulator(case):
    run_eval_case(emu_factory, case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["toy5p_l05", "gt22p_l08"])
def test_batched_eval_matches_the_reference_sampler_on_gpu(case):
    from maro_amd.cim.engine import CimBatchEngine
    run_eval_case(lambda topo, n, **kw: CimBatchEngine(topo, n, **kw), case, n_envs=3)


@pytest.mark.gpu
def test_eval_with_the_fused_dqn_equals_a_greedy_loop_on_gpu():
    """eval(FusedPerPortDQN) = reset, then act (epsilon ignored: exploit) -> step until every env is done."""
    from maro_amd.cim.engine import CimBatchEngine
    from maro_amd.cim.policy import FusedPerPortDQN, random_chains
    n, dur = 48, 180
    seeds = torch.arange(n, dtype=torch.int64) + 900
    eng = CimBatchEngine("toy.5p_ssddd_l0.5", n, durations=dur, max_actions=1, max_snapshots=16)
    smp = CimBatchSampler(eng)
    actor = FusedPerPortDQN(eng, random_chains(5, smp.state_dim, 21, seed=3), epsilon=0.5)
    res = smp.eval(actor, num_episodes=2, seeds=lambda ep: seeds + 1000 * ep)
    assert actor._m.epsilon == 0.5
    greedy = FusedPerPortDQN(eng, random_chains(5, smp.state_dim, 21, seed=3), epsilon=0.0)
    acts = torch.zeros((n, 1, 4), dtype=torch.int32, device="cuda"); nact = torch.zeros(n, dtype=torch.int32, device="cuda")
    for ep in range(2):
        eng.reset(seeds + 1000 * ep)
        eng.step()
        while not bool(eng.done.all()):
            greedy.act(acts, nact)
            eng.step(acts, nact)
        assert torch.equal(res["info"][ep]["env_metric"], eng.metrics), ep
    assert not torch.equal(res["info"][0]["env_metric"], res["info"][1]["env_metric"])
