"""Parity of the EXACT configuration bench.py times (VERDICT r01 "next" item 2): the engines bench.build_cim_groups() creates —
plan-specialised kernels, order table, fused observation, snapshot ring, G groups on their own streams, the launch form in
effect — run one complete episode with the device agent while a sample of envs is recorded step by step, and every recorded
env is replayed on the CPU oracle: each decision payload, metric triple, done flag, the fused observation (every few steps)
and the final snapshot ring must be identical.  Used by tests/test_gpu_bench_parity.py and by bench.py's untimed "parity"
leg.  TEST INFRASTRUCTURE: the oracle is the checker here, never the thing measured."""
import numpy as np

PORT_Q = ["empty", "full", "on_shipper", "on_consignee", "booking", "shortage", "fulfillment"]   # bench.QUERY_ATTRS
VESSEL_Q = ["empty", "full", "remaining_space"]                                                  # bench.VESSEL_QUERY_ATTRS


def replay_against_oracle(engines, bufs, streams, sizes, offs, seed_base, topology, durations, k, obs=True, obs_every=8):
    import torch

    from maro_amd.cim.engine import MATRIX_ATTRS, PORT_ATTRS, VESSEL_ATTRS
    from oracle.cim_oracle import CimOracle, hash_policy_action
    G, dev = len(engines), engines[0].device
    per = max(1, (k + G - 1) // G)
    picks = [sorted({int(x) for x in np.linspace(0, sizes[g] - 1, per)}) for g in range(G)]
    idx = [torch.tensor(picks[g], dtype=torch.int64, device=dev) for g in range(G)]
    cap = 4 * durations + 64
    P, S = engines[0].layout.n_ports, engines[0].layout.ring_slots
    rec = [dict(dec=torch.zeros((cap, len(picks[g]), 8), dtype=torch.int32, device=dev),
                met=torch.zeros((cap, len(picks[g]), 3), dtype=torch.int64, device=dev),
                done=torch.zeros((cap, len(picks[g])), dtype=torch.uint8, device=dev),
                op=torch.zeros((cap // obs_every + 1, len(picks[g]), P * len(PORT_Q)), dtype=torch.float64, device=dev) if obs else None,
                ov=torch.zeros((cap // obs_every + 1, len(picks[g]), len(VESSEL_Q)), dtype=torch.float64, device=dev) if obs else None)
           for g in range(G)]

    torch.cuda.synchronize(dev)   # (index / record tensors were created on torch's default stream; the groups use their own)

    def record(i):
        for g, eng in enumerate(engines):
            with torch.cuda.stream(streams[g]):
                rec[g]["dec"][i] = eng.decisions[idx[g]]
                rec[g]["met"][i] = eng.metrics[idx[g]]
                rec[g]["done"][i] = eng.done[idx[g]]
                if obs and i % obs_every == 0:
                    rec[g]["op"][i // obs_every] = bufs[g]["obs"][0][idx[g]].reshape(len(picks[g]), -1)
                    rec[g]["ov"][i // obs_every] = bufs[g]["obs"][1][idx[g]]

    # engines whose agent is answered inside the step kernel (mrx_cim_set_device_agent) are replayed in exactly that form
    fused = [getattr(eng, "_agent_keep", (None,))[0] is not None for eng in engines]
    for g, eng in enumerate(engines):
        eng.reset(torch.arange(sizes[g], dtype=torch.int64) + seed_base + offs[g] + 1)
        if fused[g]:
            eng.set_device_agent(bufs[g]["actions"], bufs[g]["n_actions"], bufs[g].get("counts"), next_key=1)
        eng.step()
    record(0)
    i = 0
    while i < cap - 1:
        i += 1
        for g, eng in enumerate(engines):
            if not fused[g]:
                eng.random_policy(i, bufs[g]["actions"], bufs[g]["n_actions"], None)
            eng.step(bufs[g]["actions"], bufs[g]["n_actions"])
        record(i)
        if i % 128 == 0:
            torch.cuda.synchronize(dev)
            if all(bool(e.done.all().item()) for e in engines):
                break
    torch.cuda.synchronize(dev)
    n_steps = i + 1
    # the final snapshot ring of the sampled envs (every attribute of every node type)
    last = [max(0, durations - S + j) for j in range(min(S, durations))]
    fis = torch.tensor(last, dtype=torch.int32, device=dev)
    ring = []
    for g, eng in enumerate(engines):
        with torch.cuda.stream(streams[g]):
            qp = eng.query("ports", fis, torch.arange(P, dtype=torch.int32, device=dev), PORT_ATTRS)[idx[g]].cpu().numpy()
            qv = eng.query("vessels", fis, torch.arange(eng.layout.n_vessels, dtype=torch.int32, device=dev), VESSEL_ATTRS)[idx[g]].cpu().numpy()
            qm = eng.query("matrices", fis, torch.zeros(1, dtype=torch.int32, device=dev), MATRIX_ATTRS)[idx[g]].cpu().numpy()
        ring.append((qp, qv, qm))
    host = [{key: (v[:n_steps].cpu().numpy() if key in ("dec", "met", "done") else (None if v is None else v.cpu().numpy())) for key, v in r.items()} for r in rec]
    status_bad = sum(int((e.status != 0).sum().item()) for e in engines)

    checked = steps_checked = obs_checks = 0
    first = None
    V = engines[0].layout.n_vessels
    for g in range(G):
        for j, e in enumerate(picks[g]):
            seed = seed_base + offs[g] + e + 1
            o = CimOracle(topology, durations=durations, max_snapshots=S)
            o.set_seed(seed)
            o.reset(keep_seed=True)
            met, dec, done = o.step(None)
            for t in range(n_steps):
                gd, gm, gdn = host[g]["dec"][t, j], host[g]["met"][t, j], bool(host[g]["done"][t, j])
                ok = gdn == done and (done or (np.array_equal(gd, dec) and np.array_equal(gm, met)))
                if ok and obs and not done and t % obs_every == 0:
                    op = o.query("ports", [int(dec[6])], list(range(P)), PORT_Q)
                    ov = o.query("vessels", [int(dec[6])], [int(dec[2])], VESSEL_Q)
                    ok = np.array_equal(host[g]["op"][t // obs_every, j], op) and np.array_equal(host[g]["ov"][t // obs_every, j], ov)
                    obs_checks += 1
                if not ok:
                    first = first or dict(group=g, env=e, step=t, gpu=[gd.tolist(), gm.tolist(), gdn], oracle=[dec.tolist(), met.tolist(), done])
                    break
                steps_checked += 1
                if done:
                    break
                met, dec, done = o.step([hash_policy_action(seed, t + 1, dec)])
            else:
                first = first or dict(group=g, env=e, step=n_steps, error="episode did not finish inside the recording")
            if first is None:   # the final ring
                qp, qv, qm = ring[g]
                same = (np.array_equal(qp[j].reshape(-1), o.query("ports", last, list(range(P)), PORT_ATTRS))
                        and np.array_equal(qv[j].reshape(-1), o.query("vessels", last, list(range(V)), VESSEL_ATTRS))
                        and np.array_equal(qm[j].reshape(-1), o.query("matrices", last, [0], MATRIX_ATTRS)))
                if not same:
                    first = dict(group=g, env=e, step="final ring")
            checked += 1
            if first is not None:
                break
        if first is not None:
            break
    return {"envs_checked": checked, "ok": first is None and status_bad == 0, "env_steps_checked": steps_checked, "observation_checks": obs_checks,
            "final_ring_frames": last, "env_status_errors": status_bad, "first_mismatch": first,
            "what": "every decision, metric, done flag, fused observation sample and the final snapshot ring of the sampled envs vs the CPU oracle "
                    "(oracle/cim_oracle.c), one complete episode of the timed configuration"}


def replay_citi_bike_against_oracle(eng, seeds, k=6, steps=600, obs_attrs=None, obs_every=16, obs_buf=None, scope_rows=False):
    """Parity of the citi_bike configuration bench.py times (BASELINE config 4): the SAME engine (plan-specialised kernels, batch size,
    ring) is reset and stepped `steps` times with the device policy while `k` sampled envs are recorded on the device; each is then
    replayed on the pure-Python oracle: every decision event, action scope, metric triple, done flag, the policy's action (against
    its Python twin) and — every `obs_every` steps — the stations observation slice the bench loop queries."""
    import torch

    from maro_amd.citi_bike.abi import draw_transfer_times
    from oracle.citi_bike_oracle import CitiBikeOracle
    from tests.cb_batch_check import policy_action
    n, dev, S = eng.n_envs, eng.device, eng.data.n_stations
    picks = sorted({int(x) for x in np.linspace(0, n - 1, k)})
    idx = torch.tensor(picks, dtype=torch.int64, device=dev)
    cap = eng.layout.scope_cap
    rec = dict(dec=torch.zeros((steps + 1, len(picks), 8), dtype=torch.int32, device=dev), scope=torch.zeros((steps + 1, len(picks), cap, 2), dtype=torch.int32, device=dev),
               met=torch.zeros((steps + 1, len(picks), 3), dtype=torch.int64, device=dev), done=torch.zeros((steps + 1, len(picks)), dtype=torch.uint8, device=dev),
               act=torch.zeros((steps + 1, len(picks), 3), dtype=torch.int32, device=dev), nact=torch.zeros((steps + 1, len(picks)), dtype=torch.int32, device=dev))
    obs = None
    if obs_attrs:   # rows: every station, or (scope_rows: city-size plans) the stations of the decision's action scope
        obs = torch.zeros((steps // obs_every + 1, len(picks), cap if scope_rows else S, len(obs_attrs)), dtype=torch.float64, device=dev)
        stations = torch.arange(S, dtype=torch.int32, device=dev)
    actions = torch.zeros((n, 1, 3), dtype=torch.int32, device=dev)
    n_actions = torch.zeros((n,), dtype=torch.int32, device=dev)
    eng.reset(seeds=seeds)   # (the engine keeps the step budget / replay period the bench timed it with: rows that say "no decision yet" are skipped below)

    def record(i):
        rec["dec"][i], rec["scope"][i], rec["met"][i], rec["done"][i] = eng.decisions[idx], eng.scope[idx], eng.metrics[idx], eng.done[idx]
        if obs is not None and i % obs_every == 0:   # the slice the bench loop reads: the fused buffer (mrx_cb_set_observation) or the query
            nodes = eng.scope[:, :, 0].contiguous() if scope_rows else stations
            obs[i // obs_every] = obs_buf[idx] if obs_buf is not None else eng.query("stations", eng.decisions[:, 3:4], nodes, obs_attrs)[idx, 0]
    eng.step()
    record(0)
    for i in range(1, steps + 1):
        eng.random_policy(i, actions, n_actions, None)
        rec["act"][i], rec["nact"][i] = actions[idx, 0], n_actions[idx]
        eng.step(actions, n_actions)
        record(i)
    torch.cuda.synchronize(dev)
    status_bad = int((eng.status != 0).sum().item())
    host = {key: v.cpu().numpy() for key, v in rec.items()}
    hobs = None if obs is None else obs.cpu().numpy()
    tts = draw_transfer_times(eng.data, np.asarray(seeds)[picks], eng.layout.transfer_times_cap)
    first, checked, steps_checked, obs_checks, unready = None, 0, 0, 0, 0
    for j, e in enumerate(picks):
        o = CitiBikeOracle(eng.data, start_tick=eng.start_tick, durations=eng.durations, snapshot_resolution=eng.snapshot_resolution,
                           max_snapshots=eng.layout.ring_slots, transfer_times=tts[j])
        m, de, od = o.step(None)
        for i in range(steps + 1):
            d, sc = host["dec"][i, j], host["scope"][i, j]
            if d[5] == 0 and not host["done"][i, j]:   # bounded steps / a deferred env: "no decision yet" — nothing to compare, no action may follow
                if not (d[1] == -1 and d[4] == 0 and (i == steps or int(host["nact"][i + 1, j]) == 0)):
                    first = first or dict(env=e, step=i, error="a row without a decision is malformed, or was answered", gpu=d.tolist())
                    break
                unready += 1
                continue
            ok = bool(host["done"][i, j]) == od and host["met"][i, j].tolist() == [m["trip_requirements"], m["bike_shortage"], m["operation_number"]]
            if ok and not od:
                ok = d[:6].tolist() == [de["tick"], de["station_idx"], de["type"], de["frame_index"], len(de["action_scope"]), 1] and \
                    [tuple(x) for x in sc[: d[4]].tolist()] == [tuple(x) for x in de["action_scope"]]
                if ok and hobs is not None and i % obs_every == 0:
                    if scope_rows:   # row r = station of scope row r; padding rows are zeros
                        nodes = [x[0] for x in de["action_scope"]]
                        want = np.zeros((cap, len(obs_attrs)))
                        want[:len(nodes)] = o.query("stations", [de["frame_index"]], nodes, list(obs_attrs)).reshape(len(nodes), -1)
                        ok = np.array_equal(hobs[i // obs_every, j], want)
                    else:
                        ok = np.array_equal(hobs[i // obs_every, j].reshape(-1), o.query("stations", [de["frame_index"]], list(range(S)), list(obs_attrs)))
                    obs_checks += 1
            if not ok:
                first = first or dict(env=e, step=i, gpu=[d.tolist(), host["met"][i, j].tolist(), bool(host["done"][i, j])], oracle=[de, dict(m), od])
                break
            steps_checked += 1
            if od or i == steps:
                break
            act = policy_action(i + 1, e, de)
            ga, gn = host["act"][i + 1, j], int(host["nact"][i + 1, j])
            if not ((act is None and gn == 0) or (gn == 1 and ga.tolist() == list(act))):
                first = first or dict(env=e, step=i + 1, error="device policy action differs from its Python twin", gpu=[ga.tolist(), gn], twin=act)
                break
            m, de, od = o.step([act] if act else None)
        checked += 1
        if first is not None:
            break
    return {"envs_checked": checked, "ok": first is None and status_bad == 0 and steps_checked > 0, "env_steps_checked": steps_checked, "rows_without_decision": unready,
            "observation_checks": obs_checks,
            "env_status_errors": status_bad, "first_mismatch": first,
            "what": f"every decision event, action scope, metric triple, done flag, device-policy action and sampled stations observation of {len(picks)} envs of the "
                    f"timed engine over the first {steps} batch steps of an episode vs the pure-Python oracle (oracle/citi_bike_oracle.py)"}


def replay_collect_against_oracle(samplers, actors, seeds_of, offs, topology, k=4, num_steps=384, reset_every=32, chains=None):
    """Parity of config 5's collection loop as bench.py times it (`sample_fused_groups` over every group's engine: fused DQN act,
    record kernel, step, emission): the samplers are restarted, one call of `num_steps` interactions runs from the start of an
    episode, and `k` sampled envs are replayed on the C oracle from what the loop itself produced —
      * the env's elements in order (emitted experiences, then the ones still in its transition cache): (tick, agent) must be the
        oracle's next decision, the 171-value sampler state must equal the state built from the ORACLE's snapshots
        (examples/cim/rl/env_sampler.py:15-31) bit for bit, the env action must equal the example's translation of the model action
        (env_sampler.py:33-64) and is then applied to the oracle;
      * every emitted reward against the 99-tick decayed sum over the oracle's own fulfillment / shortage history (float32 of a
        float64 dot product: rtol 1e-6), `next_state` against the next element's state;
      * the model action against a float32 PyTorch evaluation of the same per-port network on that state (`chains`), skipping
        near ties (the MFMA chain and torch sum in different orders)."""
    import torch

    from maro_amd.cim.policy import ACTION_SPACE, PerPortDuelingQNet, translate_actions
    from maro_amd.cim.sampler import sample_fused_groups
    from oracle.cim_oracle import CimOracle
    G = len(samplers)
    s0 = samplers[0]
    for s in samplers:   # restart: empty caches, every env at the start of its episode 0 (seeds_of[g](0))
        st = getattr(s.eng, "_bound_stream", None)
        with (torch.cuda.stream(st) if st is not None else __import__("contextlib").nullcontext()):
            s._sample_init(torch.float32)
            s.eng.status.zero_()
    res = sample_fused_groups(samplers, actors, num_steps, seeds=seeds_of, reset_every=reset_every)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    per = max(1, (k + G - 1) // G)
    net = PerPortDuelingQNet(chains, len(ACTION_SPACE)) if chains is not None else None
    decay = np.array([0.97 ** i for i in range(s0.time_window)], np.float64)
    back = list(range(s0.look_back - 1))
    first, checked, elems, rewards, pol, ties = None, 0, 0, 0, 0, 0
    status_bad = sum(int((s.eng.status != 0).sum().item()) for s in samplers)
    for g, s in enumerate(samplers):
        eng, c = s.eng, s._c
        r = {key: v.cpu().numpy() for key, v in res[g].items()}
        cache = {key: c[key].cpu().numpy() for key in ("tick", "agent", "state", "action", "env_action")}
        head, tail, cap = s._head.cpu().numpy(), s._tail.cpu().numpy(), s._cap
        seeds = seeds_of[g](torch.zeros(eng.n_envs, dtype=torch.int64)).cpu().numpy()
        for e in sorted({int(x) for x in np.linspace(0, eng.n_envs - 1, per)}):
            if int(s._ep_env[e]) != 1:
                first = dict(group=g, env=e, error=f"the env started {int(s._ep_env[e])} episodes inside the parity call (only a call inside episode 0 can be replayed: fewer steps)")
                break
            sel = np.flatnonzero(r["env_id"] == e)
            slots = [(q & (cap - 1)) for q in range(int(tail[e]), int(head[e]))]
            seq = [dict(tick=int(r["tick"][i]), agent=int(r["agent"][i]), state=r["state"][i], action=int(r["action"][i]), env_action=r["env_action"][i],
                        reward=float(r["reward"][i]), next_state=r["next_state"][i], emitted=True) for i in sel]
            seq += [dict(tick=int(cache["tick"][e, j]), agent=int(cache["agent"][e, j]), state=cache["state"][e, j], action=int(cache["action"][e, j]),
                         env_action=cache["env_action"][e, j], emitted=False) for j in slots]
            o = CimOracle(topology, durations=eng.durations)
            o.set_seed(int(seeds[e]))
            o.reset(keep_seed=True)
            met, dec, done = o.step(None)
            for i, el in enumerate(seq):
                if done:
                    break    # (an env that finished inside the call starts episode 1 afterwards: only episode 0 is replayed)
                tick, port, vessel = int(dec[0]), int(dec[1]), int(dec[2])
                fut = o.query("vessels", [tick], [vessel], ["future_stop_list"]).astype(np.int32).tolist()
                want = np.concatenate([o.query("ports", [max(0, tick - b) for b in back], [port] + fut, s.port_attributes),
                                       o.query("vessels", [tick], [vessel], s.vessel_attributes)]).astype(np.float32)
                ok = (el["tick"], el["agent"]) == (tick, port) and np.array_equal(el["state"], want)
                if ok:
                    wa = translate_actions(torch.tensor([el["action"]]), torch.from_numpy(dec[None, :].copy()), torch.tensor([float(want[-1])], dtype=torch.float64),
                                           torch.tensor([int(dec[5])]))[0, 0].numpy()
                    ok = np.array_equal(wa, el["env_action"])
                if ok and i > 0 and seq[i - 1]["emitted"] and "next_state" in seq[i - 1]:
                    ok = np.array_equal(seq[i - 1]["next_state"], el["state"])
                if not ok:
                    first = first or dict(group=g, env=e, element=i, got=[el["tick"], el["agent"], el["env_action"].tolist()], oracle=dec.tolist(),
                                          state_equal=bool(np.array_equal(el["state"], want)))
                    break
                if net is not None:
                    q = net(torch.from_numpy(want[None, :]), torch.tensor([port]))[0]
                    top = q.topk(2).values
                    if float(top[0] - top[1]) > 1e-4 * max(1.0, float(q.abs().max())):
                        pol += 1
                        if int(q.argmax()) != el["action"]:
                            first = first or dict(group=g, env=e, element=i, error="model action differs from the float32 evaluation", got=el["action"], want=int(q.argmax()))
                            break
                    else:
                        ties += 1
                elems += 1
                met, dec, done = o.step([tuple(int(x) for x in el["env_action"])])
            if first is None:    # delayed rewards of the emitted elements, from the oracle's own history
                now = o.tick
                for el in seq:
                    if not el["emitted"] or el["tick"] + s.time_window > now:
                        continue
                    h = o.query("ports", list(range(el["tick"] + 1, el["tick"] + 1 + s.time_window)), [el["agent"]], ["fulfillment", "shortage"]).reshape(-1, 2)
                    want_r = np.float32(s.ff * (h[:, 0] @ decay) - s.sf * (h[:, 1] @ decay))
                    if not np.isclose(el["reward"], want_r, rtol=1e-6, atol=1e-6):
                        first = first or dict(group=g, env=e, error="reward", tick=el["tick"], got=el["reward"], want=float(want_r))
                        break
                    rewards += 1
            checked += 1
            if first is not None:
                break
        if first is not None:
            break
    return {"envs_checked": checked, "ok": first is None and status_bad == 0 and elems > 0, "elements_checked": elems, "rewards_checked": rewards,
            "policy_choices_checked": pol, "policy_near_ties_skipped": ties, "env_status_errors": status_bad, "first_mismatch": first,
            "what": f"one sample_fused call of {num_steps} interactions from the start of an episode; per sampled env every element's (tick, agent), sampler state "
                    "(bit-exact vs the state built from the C oracle's snapshots), env action (vs the example's translation) and emitted delayed reward "
                    "(rtol 1e-6), with the oracle driven by the loop's own env actions; model actions vs a float32 PyTorch evaluation of the same network"}
