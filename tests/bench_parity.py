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

    for g, eng in enumerate(engines):
        eng.reset(torch.arange(sizes[g], dtype=torch.int64) + seed_base + offs[g] + 1)
        eng.step()
    record(0)
    i = 0
    while i < cap - 1:
        i += 1
        for g, eng in enumerate(engines):
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
