"""Shared body of the citi_bike batch-vs-oracle parity tests (CPU harness and GPU): every env of a batch gets its
own transfer-time stream and its own counter-based actions; each is replayed through the pure-Python oracle."""
import numpy as np

from maro_amd.citi_bike.abi import NODE_TYPE, STATION_ATTRS, draw_transfer_times
from oracle.citi_bike_oracle import CitiBikeOracle

M64 = (1 << 64) - 1


def mix64(x):
    x = (x + 0x9E3779B97F4A7C15) & M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M64
    return x ^ (x >> 31)


def policy_action(step, env, de):
    """Python twin of cb::random_policy_env (maro_amd/csrc/cb_device.h)."""
    scope = de["action_scope"]
    if len(scope) < 2:
        return None
    (self_s, self_max), (other, other_max) = scope[-1], scope[0]
    m = max(min(self_max, other_max), 0)
    number = mix64(mix64(step) ^ env) % (m + 1)
    return (self_s, other, number) if de["type"] == 0 else (other, self_s, number)


def run_batch_vs_oracle(backend, data, kwargs, seeds, episodes=1, check_envs=None):
    n = backend.n_envs
    tts = draw_transfer_times(data, seeds, backend.layout.transfer_times_cap)
    check_envs = list(range(n)) if check_envs is None else check_envs
    for ep in range(episodes):
        backend.reset(transfer_times=tts if ep == 0 else None)
        oracles = {e: CitiBikeOracle(data, transfer_times=tts[e], **kwargs) for e in check_envs}
        o_out = {e: o.step(None) for e, o in oracles.items()}
        dec, scope, met, done = backend.step()
        step = 0
        while True:
            for e in check_envs:
                m, de, od = o_out[e]
                assert bool(done[e]) == od, (e, step)
                assert met[e].tolist() == [m["trip_requirements"], m["bike_shortage"], m["operation_number"]], (e, step)
                if not od:
                    assert dec[e, :6].tolist() == [de["tick"], de["station_idx"], de["type"], de["frame_index"], len(de["action_scope"]), 1], (e, step, de, dec[e])
                    assert [tuple(x) for x in scope[e, : dec[e, 4]].tolist()] == [tuple(x) for x in de["action_scope"]], (e, step)
            if done.all():
                break
            step += 1
            a, na = backend.random_policy(dec, scope, step)
            for e in check_envs:
                m, de, od = o_out[e]
                if od:
                    continue
                act = policy_action(step, e, de)
                assert (act is None and na[e] == 0) or (na[e] == 1 and a[e, 0].tolist() == list(act)), (e, step, act, a[e])
                o_out[e] = oracles[e].step([act] if act else None)
            dec, scope, met, done = backend.step(a, na)
        # full-history snapshot tensors
        for e in check_envs:
            o = oracles[e]
            fis = o.frame_indices()
            S = data.n_stations
            got = backend.query(NODE_TYPE["stations"], fis, list(range(S)), list(range(len(STATION_ATTRS))), len(STATION_ATTRS))[e].reshape(-1)
            assert np.array_equal(got, o.query("stations", fis, [], STATION_ATTRS)), e
            got = backend.query(NODE_TYPE["matrices"], fis, [0], [0], S * S)[e].reshape(-1)
            assert np.array_equal(got, o.query("matrices", fis, [], ["trips_adj"])), e
        assert (backend.hdr()[13, :n] == 0).all(), backend.hdr()[13, :n]
    return step


def run_bounded_vs_oracle(backend, data, kwargs, seeds, budget, check_envs=None, max_calls=200000):
    """Bounded steps (mrx_cb_set_step_budget): a step call may leave an env without a decision (valid = 0); every env must
    still go through exactly the oracle's sequence of decisions / metrics, and end with the same snapshots.  The policy acts
    per env on that env's own decision counter, so the grouping into calls cannot leak into the trajectories."""
    n = backend.n_envs
    tts = draw_transfer_times(data, seeds, backend.layout.transfer_times_cap)
    check_envs = list(range(n)) if check_envs is None else check_envs
    backend.reset(transfer_times=tts)
    backend.set_step_budget(budget)
    oracles = {e: CitiBikeOracle(data, transfer_times=tts[e], **kwargs) for e in check_envs}
    o_out = {e: o.step(None) for e, o in oracles.items()}
    n_dec = np.zeros(n, np.int64)
    a = np.zeros((n, backend.max_actions, 3), np.int32)
    na = np.zeros(n, np.int32)
    unready = calls = 0
    dec, scope, met, done = backend.step()
    while True:
        calls += 1
        assert calls < max_calls
        a[:], na[:] = 0, 0
        for e in range(n):
            if done[e]:
                continue
            if not dec[e, 5]:
                unready += 1
                assert dec[e, 1] == -1 and dec[e, 4] == 0
                continue
            n_dec[e] += 1
            de = dict(tick=int(dec[e, 0]), station_idx=int(dec[e, 1]), type=int(dec[e, 2]), frame_index=int(dec[e, 3]),
                      action_scope=[tuple(x) for x in scope[e, : dec[e, 4]].tolist()])
            act = policy_action(int(n_dec[e]), e, de)
            if act:
                a[e, 0], na[e] = act, 1
            if e in oracles:
                m, ode, od = o_out[e]
                assert not od, (e, calls)
                assert met[e].tolist() == [m["trip_requirements"], m["bike_shortage"], m["operation_number"]], (e, calls)
                assert (de["tick"], de["station_idx"], de["type"], de["frame_index"]) == (ode["tick"], ode["station_idx"], ode["type"], ode["frame_index"]), (e, calls, de, ode)
                assert de["action_scope"] == [tuple(x) for x in ode["action_scope"]], (e, calls)
                o_out[e] = oracles[e].step([act] if act else None)
        if done.all():
            break
        dec, scope, met, done = backend.step(a, na)
    for e in check_envs:
        o = oracles[e]
        m, _, od = o_out[e]
        assert od and met[e].tolist() == [m["trip_requirements"], m["bike_shortage"], m["operation_number"]], e
        fis = o.frame_indices()
        S = data.n_stations
        got = backend.query(NODE_TYPE["stations"], fis, list(range(S)), list(range(len(STATION_ATTRS))), len(STATION_ATTRS))[e].reshape(-1)
        assert np.array_equal(got, o.query("stations", fis, [], STATION_ATTRS)), e
    assert (backend.hdr()[13, :n] == 0).all()
    backend.set_step_budget(0)
    return calls, unready


def run_joint_vs_oracle(backend, data, kwargs, seeds, mode, budget=0, check_envs=None, max_calls=100000):
    """Joint (1) / JointWithSequentialAction (2) batches, every env with its own transfer-time stream and its own counter-based
    choice of how many of the reported events it answers; optional bounded steps on top.  Every env must follow the oracle's
    step_joint: the same events, scopes (evaluated at report time on both sides), metrics, and final snapshots."""
    n, S = backend.n_envs, data.n_stations
    tts = draw_transfer_times(data, seeds, backend.layout.transfer_times_cap)
    check_envs = list(range(n)) if check_envs is None else check_envs
    backend.reset(transfer_times=tts)
    backend.set_step_budget(budget)
    oracles = {e: CitiBikeOracle(data, transfer_times=tts[e], **kwargs) for e in check_envs}
    o_out = {e: o.step_joint(None, mode) for e, o in oracles.items()}
    n_rep = np.zeros(n, np.int64)
    a = np.full((n, S, backend.max_actions, 3), -1, np.int32)
    na = np.zeros((n, S), np.int32)
    nans = np.zeros(n, np.int32)
    calls = events = 0
    dec, scope, met, done = backend.step_joint()
    while True:
        calls += 1
        assert calls < max_calls
        a[:], na[:], nans[:] = -1, 0, 0
        for e in range(n):
            if done[e] or not dec[e, 0, 5]:
                continue                      # finished, or (bounded steps) no decision yet
            n_ev = int(dec[e, 0, 6])
            n_rep[e] += 1
            r = mix64(mix64(int(n_rep[e])) ^ e)
            k = n_ev if r % 3 == 0 else int(r >> 8) % (n_ev + 1)
            if mode == 2 and k == 0:
                k = 1                           # (the reference would report the same events forever)
            des = [dict(tick=int(dec[e, i, 0]), station_idx=int(dec[e, i, 1]), type=int(dec[e, i, 2]), frame_index=int(dec[e, i, 3]),
                        action_scope=[tuple(x) for x in scope[e, i, : dec[e, i, 4]].tolist()]) for i in range(n_ev)]
            acts = [policy_action(int(n_rep[e]) * 64 + i, e, de) for i, de in enumerate(des)]
            for i in range(k):
                if acts[i]:
                    a[e, i, 0], na[e, i] = acts[i], 1
            nans[e] = k
            events += n_ev
            if e in oracles:
                m, odes, od = o_out[e]
                assert not od and len(odes) == n_ev, (e, calls, n_ev, None if odes is None else len(odes))
                assert met[e].tolist() == [m["trip_requirements"], m["bike_shortage"], m["operation_number"]], (e, calls)
                for de, ode in zip(des, odes):
                    assert (de["tick"], de["station_idx"], de["type"], de["frame_index"]) == (ode["tick"], ode["station_idx"], ode["type"], ode["frame_index"]), (e, calls, de, ode)
                    assert de["action_scope"] == [tuple(x) for x in ode["action_scope"]], (e, calls, de, ode)
                o_out[e] = oracles[e].step_joint([[x] if x else None for x in acts[:k]], mode)
        if done.all():
            break
        dec, scope, met, done = backend.step_joint(a, na, nans)
    for e in check_envs:
        o = oracles[e]
        m, _, od = o_out[e]
        assert od and met[e].tolist() == [m["trip_requirements"], m["bike_shortage"], m["operation_number"]], e
        fis = o.frame_indices()
        got = backend.query(NODE_TYPE["stations"], fis, list(range(S)), list(range(len(STATION_ATTRS))), len(STATION_ATTRS))[e].reshape(-1)
        assert np.array_equal(got, o.query("stations", fis, [], STATION_ATTRS)), e
    assert (backend.hdr()[13, :n] == 0).all()
    backend.set_step_budget(0)
    return calls, events
