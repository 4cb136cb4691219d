#!/usr/bin/env python3
"""bench.py — env-steps/sec of the batched CIM rollout engine on MI355X.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on):
    CIM global_trade.22p_l0.8, 16384 envs per GPU, durations 1120, device-side random legal agent,
    snapshot_list["ports"][frame::7 attrs] and ["vessels"][frame:vessel:3 attrs] sliced for every env every step.
One "step" = one pass of the hot path over the whole batch:
    random-policy kernel -> mrx_cim_step (action + ticks until the next decision) -> snapshot query.
value = decision events resolved per second, whole job (all ranks), state resident in HBM.

    python bench.py --gpus 1 --steps 400 --warmup 100
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
# (--groups above 3: the HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues, default 4 — three groups
# plus torch's own stream; a fourth group shares a queue and its kernels serialise: 178 M env-steps/s, against 286 M with
# GPU_MAX_HW_QUEUES=8 in the environment.  Three groups on the default queues is what the default run uses: profiles/r03_lds_diet.md §5.)

# reference-dtype frame bytes per env and targets per topology family
FRAME_BYTES = {"global_trade.22p": 15412, "toy.4p_ssdd": 886}
QUERY_ATTRS = ["empty", "full", "on_shipper", "on_consignee", "booking", "shortage", "fulfillment"]
VESSEL_QUERY_ATTRS = ["empty", "full", "remaining_space"]
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def init_dist(world, local_rank, build=None):
    """One process per GPU over RCCL (backend "nccl").  Rank 0 runs `build()` (compiles whatever is stale: the extension, the
    plan-specialised code objects) and every other rank waits for it at a barrier BEFORE it loads anything.
    MRX_BENCH_BACKEND=gloo + MRX_BENCH_DEVICE=<i> are test hooks only: they let a 1-GPU box run the N>1 code path with
    every rank on the same device (RCCL refuses two ranks per GPU)."""
    import torch
    dev = torch.device(f"cuda:{os.environ.get('MRX_BENCH_DEVICE', local_rank)}")
    rank = int(os.environ.get("RANK", "0"))
    if world <= 1:
        if build is not None:
            build()
        return None, dev
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(dev)
    backend = os.environ.get("MRX_BENCH_BACKEND", "nccl")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    if build is not None and rank == 0:
        build()
    dist.barrier()   # ranks != 0 get here first and wait: nothing is imported from the package before the build is complete
    return dist, dev


def exchange_transport():
    """None on a real run (device tensors go to RCCL as they are).  Under the MRX_BENCH_BACKEND=gloo test hook the ranks share
    one GPU and gloo's send / recv take host tensors: the test harness's host-staging transport (tests/transport.py)."""
    if os.environ.get("MRX_BENCH_BACKEND", "nccl") == "nccl":
        return None
    from tests.transport import HostStaging
    return HostStaging()


def frame_bytes(topo):
    for k, v in FRAME_BYTES.items():
        if topo.name.startswith(k):
            return v
    P, V = topo.n_ports, topo.n_vessels
    return P * 48 + V * (9 * 4 + 2 + (2 * topo.past_stop_number + 2 * topo.future_stop_number) * 4) + 4 * (P * P + 2 * V * P)


def host_cores():
    """Host cores this process may actually use: the affinity mask capped by the cgroup CPU quota (a container that sees
    256 CPUs but is limited to 16 CPU-seconds per second runs slower with 256 busy threads than with 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())       # cgroup v1
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline(topology, durations, budget_s):
    """The C oracle (a port of the reference algorithm) timed on the host: whole episodes of the same workload with the
    same counter-based agent — first on ONE core, then one independent oracle per host core (threads; the ctypes call
    releases the GIL), ~budget_s / 2 seconds each.  `value` is the all-cores rate, `cores` the threads used."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle.cim_oracle import CimOracle

    def worker(o, wid, budget):
        t0 = time.perf_counter()
        steps, ticks, episodes = o.bench(1000 + 1000003 * wid, budget)   # the episode loop runs in C (oracle/cim_oracle.c)
        return steps, ticks, episodes, time.perf_counter() - t0

    first = CimOracle(topology, durations=durations)
    s1, k1, e1, d1 = worker(first, 0, budget_s / 2)
    cores = host_cores()
    oracles = [first] + [CimOracle(first.topo, durations=durations) for _ in range(cores - 1)]  # built outside the timed window
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        res = list(ex.map(lambda w: worker(oracles[w], w + 1, budget_s / 2), range(cores)))
    dt = time.perf_counter() - t0
    steps, ticks, episodes = (sum(r[i] for r in res) for i in range(3))
    return {"value": steps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port", "value_one_core": s1 / d1,
            "sample": f"{episodes} full episodes of {topology} ({durations} ticks, reset+rollout in C, {steps} decisions, "
                      f"{dt:.1f} s on {cores} threads = one oracle per usable host core (affinity capped by the cgroup CPU quota); {ticks / dt:.0f} ticks/s); "
                      f"one core alone: {e1} episodes, {s1} decisions in {d1:.1f} s",
            "note": "a C port of the reference algorithm (oracle/cim_oracle.c); the reference's own Env.step is in cpu_baseline_reference"}


def cpu_baseline_citi_bike(topology, durations, res, budget_s):
    """The pure-Python oracle (a port of the reference algorithm) timed on ONE host core, same agent."""
    import numpy as np

    from maro_amd.citi_bike.abi import draw_transfer_times
    from maro_amd.citi_bike.data import load_topology
    from oracle.citi_bike_oracle import CitiBikeOracle
    from tests.cb_batch_check import policy_action

    data = load_topology(topology)
    steps = episodes = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        tt = draw_transfer_times(data, [episodes], 3 * (durations // data.resolution + 1))[0]
        o = CitiBikeOracle(data, durations=durations, snapshot_resolution=res, max_snapshots=16, transfer_times=tt)
        m, de, done = o.step(None)
        k = 0
        while not done and time.perf_counter() - t0 < budget_s:
            k += 1
            act = policy_action(k, episodes, de)
            m, de, done = o.step([act] if act else None)
            steps += 1
        episodes += 1
    dt = time.perf_counter() - t0
    return {"value": steps / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"{steps} decisions of {topology} ({episodes} episode(s), pure-Python oracle, {dt:.1f} s on 1 core)"}


def cpu_baseline_citi_bike_reference(topology, durations, res, budget_s):
    """The REAL reference's citi_bike Env (maro/simulator/scenarios/citi_bike/business_engine.py:101-147) timed on THIS box, ONE
    process, same agent as the port leg — in a child interpreter on the built reference (reference_runtime).  The toy topologies'
    build folders are written back from the packaged .npz by the checker tooling (oracle/setup_toy_topologies.py: every array is
    asserted to survive the round trip) into the runtime's private HOME.  None where no built reference is reachable or the
    topology is not one of the toys."""
    rt = reference_runtime()
    if rt is None:
        return None
    try:
        from oracle.setup_toy_topologies import TOYS, ensure_toy
        if topology not in TOYS:
            return None
        ensure_toy(rt[0], rt[2], topology)
    except Exception as e:
        return {"error": f"toy write-back failed: {e}"[:300]}
    code = r"""
import sys, time
from maro.simulator import Env
from maro.simulator.scenarios.citi_bike.common import Action, DecisionType
sys.path.insert(0, sys.argv[5])
from tests.cb_batch_check import policy_action
topo, dur, res, budget = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
env = Env(scenario="citi_bike", topology=topo, start_tick=0, durations=dur, snapshot_resolution=res)
steps = episodes = 0
t_step = t_reset = 0.0
t_all = time.perf_counter()
while time.perf_counter() - t_all < budget:
    t0 = time.perf_counter()
    if episodes:
        env.reset()
    t_reset += time.perf_counter() - t0
    t0 = time.perf_counter()
    m, de, done = env.step(None)
    k = 0
    while not done and time.perf_counter() - t_all < budget:
        k += 1
        d = {"type": 0 if de.type == DecisionType.Supply else 1, "action_scope": list(de.action_scope.items())}
        a = policy_action(k, episodes, d)
        m, de, done = env.step(Action(a[0], a[1], a[2]) if a else None)
        steps += 1
    t_step += time.perf_counter() - t0
    episodes += 1
print(steps, episodes, t_step, t_reset)
"""
    try:
        out = _run_reference(code, [topology, durations, res, budget_s, REPO], budget_s * 4 + 120)
        steps, episodes, t_step, t_reset = out.stdout.split()[-4:]
        steps, episodes, t_step, t_reset = int(steps), int(episodes), float(t_step), float(t_reset)
    except Exception as e:
        return {"error": (str(e) + " " + (out.stderr[-300:] if "out" in dir() else ""))[:400]}
    return {"value": steps / t_step, "unit": "env-steps/s", "cores": 1, "kind": "reference", "value_end_to_end": steps / (t_step + t_reset),
            "sample": f"{steps} decisions of {topology} ({episodes} episode(s) of {durations} ticks, snapshot resolution {res}) on the reference's own Env "
                      f"(maro.simulator.Env(scenario='citi_bike'), single process, same agent as the GPU leg's device policy): {t_step:.1f} s of stepping + {t_reset:.1f} s of env.reset()",
            "measured": "live, in this run", "reference_from": rt[3], **box_description()}


def measured_bytes_citi_bike(topology, n, step_budget, code_key, groups=1, replay_period=1):
    """HBM bytes one batch step of this citi_bike configuration really moves (profiles/latest_pmc_citi_bike.json: separate
    --pmc FETCH_SIZE / WRITE_SIZE passes, all kernels of one batch step) -> (bytes or None, basis).  An entry of another build
    (its code_object_key differs) or another batch size is NOT used: the fraction is then null, never a formula."""
    try:
        with open(os.path.join(REPO, "profiles", "latest_pmc_citi_bike.json")) as fp:
            pmc = json.load(fp)
    except (OSError, ValueError):
        return None, "no profiles/latest_pmc_citi_bike.json"
    state = "no PMC entry of this topology / batch size / step budget / replay period"
    for ent in pmc.get("entries", []):
        if ent["topology"] == topology and ent["envs_per_launch"] == n and ent.get("step_budget", 0) == step_budget and ent.get("groups_per_gpu", 1) == groups and \
                ent.get("replay_period", 1) == replay_period:
            if ent.get("code_object_key") != code_key:
                state = f"the PMC entry is of another build (code object {ent.get('code_object_key')}, running {code_key})"
                continue
            return (2 * ent["fetch_size_kib"] + ent["write_size_kib"]) * 1024, \
                f"measured HBM bytes (PMC: 2 x FETCH_SIZE + WRITE_SIZE summed over every kernel of one batch step, {pmc['source']}; entry of git {ent.get('git_head')})"
    return None, state


def bench_citi_bike(args, dist, dev, rank, world):
    """BASELINE.json configs[3]: citi_bike toy.3s_4t, 4096 envs per GPU, shared trip table, per-env action seeds.
    One step = device policy -> mrx_cb_step (action + ticks until the next decision) -> stations snapshot slice.
    Returns the JSON object on rank 0 (None elsewhere)."""
    import numpy as np
    import torch

    from maro_amd.citi_bike.data import load_topology as load_cb
    from maro_amd.citi_bike.engine import CitiBikeBatchEngine

    n, res = args.envs, 10
    topology = args.topology if args.topology != "global_trade.22p_l0.8" else "toy.3s_4t"
    durations = args.durations if args.durations != 1120 else 44000
    durations = min(durations, len(load_cb(topology).tick_day))   # (city.180s holds two days of trips, the toys a month)
    # Independent env GROUPS on their own HIP streams (as for CIM): a batch step of this path lasts as long as its longest
    # env-step's dependent chain, whatever the batch size, until the chip fills — so G groups' chains overlap almost for free.
    G = max(1, min(args.cb_groups, n))
    sizes = [n // G + (1 if g < n % G else 0) for g in range(G)]
    offs = [sum(sizes[:g]) for g in range(G)]
    seeds = np.arange(n) + rank * n + 1
    engines, streams, bufs = [], [], []
    for g in range(G):
        kw = dict(durations=durations, snapshot_resolution=res, max_snapshots=16, max_actions=1, device=dev, seeds=seeds[offs[g]:offs[g] + sizes[g]])
        try:
            e = CitiBikeBatchEngine(topology, sizes[g], specialize=bool(args.specialize), **kw)   # kernels compiled for this plan (cached in-tree)
        except (RuntimeError, OSError, subprocess.CalledProcessError) as err:
            if not args.specialize:
                raise
            print(f"bench: specialised kernels unavailable ({err}); using the generic ones", file=sys.stderr)
            e = CitiBikeBatchEngine(topology, sizes[g], specialize=False, **kw)
        engines.append(e)
        streams.append(torch.cuda.Stream(device=dev) if G > 1 else torch.cuda.current_stream(dev))
    eng = engines[0]
    S = eng.data.n_stations
    q_attrs = ["bikes", "shortage", "trip_requirement", "fulfillment", "capacity", "extra_cost", "min_bikes"]
    # the per-step observation: on the toys every station; on city-sized topologies the stations of the decision's ACTION SCOPE (the
    # deciding station + its filtered neighbours: what an agent can act on, 21 rows with the ny filter chain) — all 800 stations x
    # 7 attributes per env and step would be 45 KB of float64 per env-step, several times the simulation's own traffic
    scope_obs = S > 64
    fused_obs = args.obs == "fused" and not args.no_query and (not scope_obs or bool(eng.specialized))   # (scope rows: the wave kernels of a specialised plan)
    cap = eng.layout.scope_cap
    stations = torch.arange(S, dtype=torch.int32, device=dev)
    for g, e in enumerate(engines):
        ng = sizes[g]
        if args.step_budget:
            e.set_step_budget(args.step_budget)
        if args.replay_overlap:
            e.set_replay_overlap(True)
        if args.replay_period > 1:   # (the groups' replay calls staggered: one group's replay kernel beside the others' in-tick kernels)
            e.set_replay_period(args.replay_period, (g * args.replay_period) // G)
        bufs.append(dict(actions=torch.zeros((ng, 1, 3), dtype=torch.int32, device=dev), n_actions=torch.zeros((ng,), dtype=torch.int32, device=dev),
                         counter=torch.zeros((1,), dtype=torch.int64, device=dev),
                         q_nodes=torch.empty((ng, cap), dtype=torch.int32, device=dev) if scope_obs else None,
                         q_out=None if (args.no_query or fused_obs) else torch.empty((ng, 1, cap if scope_obs else S, len(q_attrs)), dtype=torch.float64, device=dev)))
        if fused_obs:   # the same slice written by the step kernel itself (mrx_cb_set_observation): no query launch
            bufs[-1]["obs"] = e.set_observation(q_attrs)
    torch.cuda.synchronize(dev)
    if G > 1:
        for e, st in zip(engines, streams):
            e.use_stream(st)

    def one_step(i, count=True):
        for g, e in enumerate(engines):
            b = bufs[g]
            if i == 0:
                e.step()
                continue
            e.random_policy(i, b["actions"], b["n_actions"], b["counter"] if count else None)
            e.step(b["actions"], b["n_actions"])
            if b["q_out"] is not None:
                if scope_obs:
                    with torch.cuda.stream(streams[g]):
                        b["q_nodes"].copy_(e.scope[:, :, 0])      # per-env node lists; -1 padding reads as zeros (query semantics)
                    e.query("stations", e.decisions[:, 3:4], b["q_nodes"], q_attrs, out=b["q_out"])
                else:
                    e.query("stations", e.decisions[:, 3:4], stations, q_attrs, out=b["q_out"])

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def total(name):
        return sum(float(getattr(e, name).to(torch.int64).sum().item()) for e in engines)

    t_leg = time.perf_counter()
    step_i = 0
    for _ in range(args.warmup):
        one_step(step_i)
        step_i += 1
    sync_all()
    # the timed window: exactly --steps batch steps between barrier + synchronize, --repeats times back to back (value = median)
    windows = []
    for _ in range(max(1, args.repeats)):
        for b in bufs:
            b["counter"].zero_()
        tick0 = total("ticks")
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one_step(step_i)
            step_i += 1
        t_issued = time.perf_counter() - t0
        sync_all()
        dt = time.perf_counter() - t0
        windows.append((dt, sum(float(b["counter"].item()) for b in bufs), total("ticks") - tick0, t_issued))
    n_done = int(total("done"))
    status_bad = sum(int((e.status != 0).sum().item()) for e in engines)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(min(args.steps, 100))]
    for a, b in ev:       # (group 0's step launch alone on its stream: informational)
        eng.random_policy(step_i, bufs[0]["actions"], bufs[0]["n_actions"], None)
        a.record(streams[0])
        eng.step(bufs[0]["actions"], bufs[0]["n_actions"])
        b.record(streams[0])
        step_i += 1
    torch.cuda.synchronize(dev)
    step_kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
    # the same loop with bounded steps (mrx_cb_set_step_budget): a call no longer waits for the batch's longest env-step; envs
    # without a decision yet are skipped by the policy and continue in the next call.  Reported beside `value`, not as it.
    bounded = None
    if args.bounded_budget and not args.step_budget:
        for e in engines:
            e.set_step_budget(args.bounded_budget)
        for _ in range(args.warmup):
            one_step(step_i)
            step_i += 1
        sync_all()
        for b in bufs:
            b["counter"].zero_()
        sync_all()
        tb = time.perf_counter()
        for _ in range(args.steps):
            one_step(step_i)
            step_i += 1
        sync_all()
        dtb = time.perf_counter() - tb
        bounded = [sum(float(b["counter"].item()) for b in bufs), dtb]
        for e in engines:
            e.set_step_budget(0)
        sync_all()
    # the one exchange step of a sharded rollout: 32 steps of (decision, action, metrics, done) per env gathered to the
    # learner rank over RCCL (maro_amd/cim/rollout.py::gather_to_learner); outside the timed env-step window
    gather_ms = None
    if dist is not None:
        from maro_amd.cim.rollout import gather_to_learner
        T = 32
        traj = {"decisions": torch.zeros((T, n, 8), dtype=torch.int32, device=dev), "actions": torch.zeros((T, n, 1, 3), dtype=torch.int32, device=dev),
                "metrics": torch.zeros((T, n, 3), dtype=torch.int64, device=dev), "done": torch.zeros((T, n), dtype=torch.uint8, device=dev)}
        torch.cuda.synchronize(dev)   # (allocated on torch's current stream, written on the groups' streams)
        for k in range(T):
            one_step(step_i, count=False)
            for g, e in enumerate(engines):
                with torch.cuda.stream(streams[g]):
                    sl = slice(offs[g], offs[g] + sizes[g])
                    traj["decisions"][k, sl], traj["actions"][k, sl] = e.decisions, bufs[g]["actions"]
                    traj["metrics"][k, sl], traj["done"][k, sl] = e.metrics, e.done
            step_i += 1
        sync_all()
        tg = time.perf_counter()
        out = gather_to_learner(traj, dst=0, sizes=[n] * world, transport=exchange_transport())
        sync_all()
        gather_ms = (time.perf_counter() - tg) * 1e3
        if rank == 0:
            assert out["decisions"].shape[1] == n * world
    # parity (untimed, every rank's own engine; rank 0 reports): sampled envs of THIS engine replayed on the pure-Python oracle
    parity = None
    if args.parity_envs > 0 and rank == 0:
        from tests.bench_parity import replay_citi_bike_against_oracle
        torch.cuda.synchronize(dev)
        eng.use_stream(None)      # (group 0's engine, on torch's current stream for the replay)
        parity = replay_citi_bike_against_oracle(eng, seeds[:sizes[0]], k=min(args.parity_envs, 8), steps=min(600, max(64, durations // 4)),
                                                 obs_attrs=None if args.no_query else q_attrs, obs_buf=bufs[0].get("obs"), scope_rows=scope_obs)
    gpu_s = time.perf_counter() - t_leg
    R = len(windows)
    t_max = torch.tensor([w[0] for w in windows] + [bounded[1] if bounded else 0.0], dtype=torch.float64, device=dev)
    tot = torch.tensor([w[1] for w in windows] + [w[2] for w in windows] + [float(n_done), float(status_bad), bounded[0] if bounded else 0.0], dtype=torch.float64, device=dev)
    t_issued = sorted(w[3] for w in windows)[len(windows) // 2]
    if dist is not None:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    dts, dtb = [float(x) for x in t_max.tolist()[:R]], float(t_max[R])
    tl = [float(x) for x in tot.tolist()]
    res_w, tick_w = tl[:R], tl[R:2 * R]
    n_done, status_bad, resolved_b = tl[2 * R:]
    if rank != 0:
        return None
    vals = [res_w[i] / dts[i] for i in range(R)]
    med = sorted(range(R), key=lambda i: vals[i])[R // 2]
    dt, resolved, ticks_adv = dts[med], res_w[med], tick_w[med]
    tbar = ticks_adv / max(resolved, 1.0)
    ms_per_step = dt / args.steps * 1e3
    from maro_amd import _lib as mrx_lib
    code_key = f"{getattr(eng, 'code_object_key', None)}+{mrx_lib.source_hash('cb')}"   # step kernels' code object + the library's sources (policy / query kernels)
    traffic, basis = measured_bytes_citi_bike(topology, n, args.step_budget, code_key, G, args.replay_period)
    achieved = None if traffic is None else traffic / (ms_per_step * 1e-3) / 1e9     # per GPU: bytes of one batch step / its wall time
    out = {
        "metric": f"env-steps/sec (decision events/sec), citi_bike {topology}",
        "value": resolved / dt, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int32+f64", "data": "synthetic", "repeats": R, "value_min": min(vals), "value_max": max(vals), "gpu_seconds_total": gpu_s,
        "config": {"workload": f"citi_bike {topology}, {n} envs/GPU x {world} GPU, durations {durations}, resolution {res}, "
                               f"device policy, stations snapshot slice {'off' if args.no_query else ('every step (' + ('stations of the action scope' if scope_obs else 'all stations') + ' x 7 attrs, ' + ('fused into the step kernels)' if fused_obs else 'mrx_cb_query)'))}",
                   "envs_per_gpu": n, "groups_per_gpu": G, "host_enqueue_ms_per_step": t_issued / args.steps * 1e3, "specialized_kernels": bool(eng.specialized), "code_object_key": code_key, "wave_cooperative_step": bool(eng.set_wave_decisions(0)),
                   "env_major_state": bool(eng.layout.env_major), "ring_slots": 16, "parallelism": f"env-shard x{world} (no data-path collective); {G} independent group(s) per GPU on separate HIP streams",
                   "trajectory_gather_ms_32_steps": gather_ms, "mean_ticks_per_env_step": tbar, "envs_finished_in_window": n_done, "env_status_errors": status_bad,
                   "step_budget": args.step_budget, "replay_period": args.replay_period},
        "roofline": {"bound": "hbm", "kernel": "mrx_k_cb_step" if not eng.layout.env_major else "mrx_k_cb_step_wave + mrx_k_cb_replay_wave (+ query, policy)",
                     "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": None if achieved is None else achieved / HBM_PEAK_GBPS,
                     "traffic": traffic, "basis": basis, "kernel_ms": step_kernel_ms, "env_steps_per_launch": n,
                     "note": "latency-bound by construction (SURVEY.md 8d: ~1 KB per env-step): the batch step lasts as long as its longest env-step's dependent chain; "
                             "frac is measured bytes / wall time of a batch step and is null when no PMC record of this exact build exists"},
    }
    if bounded:
        out["bounded_steps"] = {"budget_records": args.bounded_budget, "value": resolved_b / dtb, "unit": "env-steps/s", "ms_per_call": dtb / args.steps * 1e3,
                                "decisions_per_call_per_env": resolved_b / (args.steps * n * world),
                                "what": "same loop with mrx_cb_set_step_budget: a call stops after ~budget events per env; envs without a decision yet continue in the next call (trajectories unchanged)"}
    if parity is not None:
        out["parity"] = parity
    if world == 1 and not args.no_cpu:
        ref = cpu_baseline_citi_bike_reference(topology, min(durations, 1440), res, args.cpu_seconds)
        port = cpu_baseline_citi_bike(topology, min(durations, 1440), res, args.cpu_seconds if ref is None or "error" in ref else min(args.cpu_seconds, 5.0))
        if ref is not None and "error" not in ref:
            out["cpu_baseline"] = dict(ref, port=port)       # the reference's own Env, live on this box; the Python port rides along
        else:
            out["cpu_baseline"] = port if ref is None else dict(port, reference_error=ref["error"])
    return out


class _EnvCounts:
    """The per-env answered-decision counters of the device agent (mrx_cim_set_device_agent) behind the two calls bench.py makes on a
    group's counter: zero_() and item() (= the sum)."""

    def __init__(self, t):
        self.t = t

    def zero_(self):
        self.t.zero_()

    def item(self):
        return int(self.t.sum().item())


def build_cim_groups(topology, n, G, dev, rank=0, durations=1120, ring=4, specialize=True, step_mode=0, obs="fused", policy="random", agent="launch"):
    """The per-GPU configuration bench.py times: the batch split into G independent groups (engine + HIP stream each), fused
    observation, snapshot ring.  Shared with tests/test_gpu_bench_parity.py so that the parity test runs EXACTLY this setup."""
    import torch

    from maro_amd.cim.engine import CimBatchEngine
    sizes = [n // G + (1 if g < n % G else 0) for g in range(G)]   # group sizes differ by at most one env
    offs = [sum(sizes[:g]) for g in range(G)]
    engines, streams, bufs = [], [], []
    for g in range(G):
        ng = sizes[g]
        seeds = torch.arange(ng, dtype=torch.int64) + rank * n + offs[g] + 1
        kw = dict(durations=durations, max_snapshots=max(ring, 8) if policy == "dqn" else ring, max_actions=1, device=dev, seeds=seeds,
                  step_mode=step_mode)  # dqn: the look-back window must fit the ring
        try:
            eng = CimBatchEngine(topology, ng, specialize=bool(specialize), **kw)
        except (RuntimeError, OSError, subprocess.CalledProcessError) as e:   # no hipcc and not in the cache: generic kernels
            if not specialize:
                raise
            print(f"bench: specialised kernels unavailable ({e}); using the generic ones", file=sys.stderr)
            eng = CimBatchEngine(topology, ng, specialize=False, **kw)
        engines.append(eng)
        streams.append(torch.cuda.Stream(device=dev) if G > 1 else torch.cuda.current_stream(dev))
        P = eng.topo.n_ports
        bufs.append(dict(actions=torch.zeros((ng, 1, 4), dtype=torch.int32, device=dev),
                         n_actions=torch.zeros((ng,), dtype=torch.int32, device=dev),
                         counter=torch.zeros((1,), dtype=torch.int64, device=dev),
                         q_ports=torch.empty((ng, 1, P, len(QUERY_ATTRS)), dtype=torch.float64, device=dev) if obs == "query" else None,
                         q_vessel=torch.empty((ng, 1, 1, len(VESSEL_QUERY_ATTRS)), dtype=torch.float64, device=dev) if obs == "query" else None))
        if obs == "fused" and policy != "dqn":
            # the same two slices, written by the step kernel itself (mrx_cim_set_observation) instead of two more launches
            bufs[-1]["obs"] = eng.set_observation(QUERY_ATTRS, VESSEL_QUERY_ATTRS)
        if agent == "fused":
            # the random legal agent answered inside the step kernel (mrx_cim_set_device_agent): no policy launch at all
            bufs[-1]["counts"] = torch.zeros((ng,), dtype=torch.int32, device=dev)
            bufs[-1]["counter"] = _EnvCounts(bufs[-1]["counts"])
            eng.set_device_agent(bufs[-1]["actions"], bufs[-1]["n_actions"], bufs[-1]["counts"], next_key=1)
    for eng, st in zip(engines, streams):
        eng.use_stream(st)   # every engine call goes to its group's stream without a per-call stream switch
    return engines, streams, bufs, sizes, offs


_REF_RUNTIME = None


def reference_runtime():
    """Where the REAL reference (microsoft/maro built by oracle/build_ref.sh) can be imported from on THIS box, for the
    cpu_baseline legs only: (maro root, stubs dir, HOME, description) or None.  Order: $MARO_REFERENCE_BUILD; the runtime archive
    oracle/_ref/maro_ref.tgz (a build output of `oracle/build_ref.sh <dest> oracle/_ref` — written by __graft_entry__.build()
    where /root/reference exists, git-ignored and NOT tracked, shipped to the GPU box with the snapshot like the built .so files)
    unpacked into a private per-user cache folder; the build container's /tmp/oracle."""
    global _REF_RUNTIME
    if _REF_RUNTIME is not None:
        return _REF_RUNTIME or None

    def built(root):
        b = os.path.join(root, "maro", "backends")
        return os.path.isdir(b) and any(f.startswith("frame.") and f.endswith(".so") for f in os.listdir(b))
    import tempfile
    env_root = os.environ.get("MARO_REFERENCE_BUILD")
    tgz = os.path.join(REPO, "oracle", "_ref", "maro_ref.tgz")
    res = False
    if env_root and built(env_root):
        res = (env_root, os.path.join(os.path.dirname(env_root), "stubs"), os.path.join(os.path.dirname(env_root), "home"), f"$MARO_REFERENCE_BUILD={env_root}")
    elif os.path.exists(tgz):
        import hashlib
        import tarfile
        # Unpacked under a directory only this user can write (0700), keyed by the archive's CONTENT hash, and trusted only when
        # the completion marker written after a full extraction is there: a half-populated or foreign folder is never executed.
        h = hashlib.sha256()
        with open(tgz, "rb") as f:
            for chunk in iter(lambda: f.read(1 << 20), b""):
                h.update(chunk)
        base = os.path.join(os.path.expanduser("~"), ".cache", "maro_amd")
        try:
            os.makedirs(base, mode=0o700, exist_ok=True)
            if os.stat(base).st_uid != os.getuid() or (os.stat(base).st_mode & 0o077):
                raise OSError("cache directory is not private")
        except OSError:
            base = tempfile.mkdtemp(prefix="maro_amd_ref_")   # private by construction (0700, fresh name)
        dest = os.path.join(base, "ref_" + h.hexdigest()[:16])
        marker = os.path.join(dest, ".complete")
        if not os.path.exists(marker):
            tmp = tempfile.mkdtemp(prefix="ref_unpack_", dir=base)
            with tarfile.open(tgz) as tf:
                try:
                    tf.extractall(tmp, filter="data")   # no absolute paths, links out of the tree, devices or setuid bits
                except TypeError:                       # Python < 3.10.12 / 3.11.4: no extraction filters
                    root_real = os.path.realpath(tmp)
                    members = [m for m in tf.getmembers() if (m.isfile() or m.isdir())
                               and os.path.realpath(os.path.join(tmp, m.name)).startswith(root_real + os.sep)]
                    tf.extractall(tmp, members=members)
            os.makedirs(os.path.join(tmp, "maro_ref", "home"), exist_ok=True)
            with open(os.path.join(tmp, ".complete"), "w") as f:
                f.write(h.hexdigest())
            try:
                os.replace(tmp, dest)
            except OSError:
                import shutil
                shutil.rmtree(tmp, ignore_errors=True)   # another rank / process got there first
        root = os.path.join(dest, "maro_ref")
        if os.path.exists(marker) and built(root):
            res = (root, os.path.join(root, "stubs"), os.path.join(root, "home"), "oracle/_ref/maro_ref.tgz (the reference built by oracle/build_ref.sh, unpacked on this box)")
    elif built("/tmp/oracle/maro_src"):
        res = ("/tmp/oracle/maro_src", "/tmp/oracle/stubs", "/tmp/oracle/home", "/tmp/oracle/maro_src (oracle/build_ref.sh in the build container)")
    _REF_RUNTIME = res
    return res or None


def _run_reference(code, argv, timeout):
    """Run `code` in a child interpreter that can import the built reference (never this process: the product must not)."""
    root, stubs, home, _ = reference_runtime()
    env = dict(os.environ, HOME=home, SKIP_DEPLOYMENT="TRUE", PYTHONPATH=os.pathsep.join([root, stubs]))
    return subprocess.run([sys.executable, "-c", code] + [str(a) for a in argv], capture_output=True, text=True, timeout=timeout, env=env)


def box_description():
    import platform
    gpu = None
    try:
        import torch
        gpu = torch.cuda.get_device_name(0) if torch.cuda.is_available() else None
    except Exception:
        pass
    return {"host": platform.node(), "gpu": gpu, "host_cores_usable": host_cores(),
            "where": ("the bench box itself (GPU host: " + gpu + ")") if gpu else "a box without a GPU (build container)"}


def cpu_baseline_reference(topology, durations, budget_s):
    """The REAL reference (microsoft/maro, built by oracle/build_ref.sh) timed on THIS box's host cores: single-process Env.step
    with a random legal agent and the reference's own batch form, maro.vector_env.VectorEnv with
    one process per usable core (step 3).  None when no built reference is reachable (see reference_runtime)."""
    rt = reference_runtime()
    if rt is None:
        return None
    code = r"""
import sys, time, random
from maro.simulator import Env
from maro.simulator.scenarios.cim.common import Action, ActionType
env = Env("cim", sys.argv[1], durations=int(sys.argv[2]))
budget = float(sys.argv[3]); rng = random.Random(0)
steps = episodes = 0; t_step = 0.0; t_reset = 0.0
t0 = time.perf_counter()
while time.perf_counter() - t0 < budget:
    tr = time.perf_counter(); env.reset(); t_reset += time.perf_counter() - tr
    ts = time.perf_counter()
    m, de, done = env.step(None)
    while not done:
        sc = de.action_scope
        if rng.random() < 0.5 and sc.load > 0: a = Action(de.vessel_idx, de.port_idx, rng.randint(0, sc.load), ActionType.LOAD)
        else: a = Action(de.vessel_idx, de.port_idx, rng.randint(0, sc.discharge), ActionType.DISCHARGE)
        m, de, done = env.step(a); steps += 1
    t_step += time.perf_counter() - ts; episodes += 1
print(steps, episodes, t_step, t_reset)
"""
    try:
        out = _run_reference(code, [topology, durations, budget_s], budget_s * 4 + 120)
        steps, episodes, t_step, t_reset = out.stdout.split()[-4:]
        steps, episodes, t_step, t_reset = int(steps), int(episodes), float(t_step), float(t_reset)
    except Exception as e:   # the reference is a convenience leg, never a reason to fail the bench
        return {"error": str(e)[:200]}
    res = {"value": steps / t_step, "unit": "env-steps/s", "cores": 1, "kind": "reference",
           "value_end_to_end": steps / (t_step + t_reset), "reset_s_per_episode": t_reset / max(episodes, 1),
           "sample": f"{episodes} full episode(s) of {topology} ({durations} ticks) on the reference's own Env (maro.simulator.Env, single process, "
                     f"Python {sys.version_info.major}.{sys.version_info.minor}): {steps} decisions in {t_step:.1f} s of stepping + {t_reset:.1f} s of env.reset()",
           "measured": "live, in this run", "reference_from": rt[3], **box_description()}
    # the reference's own batch form, one process per env on every usable core, broadcast action=None
    cores = max(1, min(host_cores(), 16))
    vcode = r"""
import sys, time
from maro.vector_env import VectorEnv
n = int(sys.argv[3])
with VectorEnv(batch_num=n, scenario="cim", topology=sys.argv[1], durations=int(sys.argv[2])) as env:
    t0 = time.perf_counter(); steps = 0
    metrics, des, done = env.step(None)
    while not done and time.perf_counter() - t0 < float(sys.argv[4]):
        steps += sum(1 for d in des if d is not None)
        metrics, des, done = env.step(None)
    print(steps, time.perf_counter() - t0)
"""
    try:
        out = _run_reference(vcode, [topology, durations, cores, budget_s], budget_s * 4 + 180)
        vs, vt = out.stdout.split()[-2:]
        res["vector_env"] = {"value": int(vs) / float(vt), "unit": "env-steps/s", "processes": cores,
                             "sample": f"maro.vector_env.VectorEnv(batch_num={cores}), action=None broadcast, {int(vs)} decisions in {float(vt):.1f} s"}
    except Exception as e:
        res["vector_env"] = {"error": str(e)[:200]}
    return res


DQN_FLOPS_PER_ENV = 2.0 * sum(a * b for a, b in ((171, 256), (256, 128), (128, 64), (64, 32), (32, 128), (32, 128), (128, 21), (128, 1)))   # the example's real layer sizes
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32 MFMA (v_mfma_f32_16x16x4_f32)


def measured_bytes_collect(n, G, code_key):
    """HBM bytes one batch interaction of the collection loop really moves (profiles/latest_pmc_collect.json: separate --pmc
    FETCH_SIZE / WRITE_SIZE passes, summed over EVERY kernel of the loop / interactions) -> (bytes or None, basis)."""
    try:
        with open(os.path.join(REPO, "profiles", "latest_pmc_collect.json")) as fp:
            pmc = json.load(fp)
    except (OSError, ValueError):
        return None, "no profiles/latest_pmc_collect.json"
    state = "no PMC entry of this batch size / group count"
    for ent in pmc.get("entries", []):
        if ent["envs_per_gpu"] == n and ent["groups_per_gpu"] == G:
            if ent.get("code_object_key") != code_key or code_key is None:
                state = f"the PMC entry is of another build (code object {ent.get('code_object_key')}, running {code_key})"
                continue
            return (2 * ent["fetch_size_kib"] + ent["write_size_kib"]) * 1024, \
                f"measured HBM bytes (PMC: 2 x FETCH_SIZE + WRITE_SIZE summed over every kernel of the loop / batch interactions, {pmc['source']}; entry of git {ent.get('git_head')})"
    return None, state


def cpu_baseline_collect(topology, durations, budget_s):
    """Config 5 on the host cores, the REAL reference: maro.rl's own CIMEnvSampler.sample(num_steps) (examples/cim/rl/env_sampler.py
    over maro/rl/rollout/env_sampler.py) on the reference Env with the example's 22 random-init DQN policies (PyTorch CPU) — timed in
    a child interpreter on the built reference (reference_runtime).  None where no built reference is reachable."""
    if reference_runtime() is None:
        return None
    code = r"""
import sys, time
from unittest.mock import MagicMock
for name in ["zmq", "zmq.asyncio", "zmq.eventloop", "zmq.eventloop.zmqstream", "tornado", "tornado.ioloop"]:
    sys.modules[name] = MagicMock()
import torch
torch.set_num_threads(1)
from maro.simulator import Env
topo, dur, budget = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
from examples.cim.rl.algorithms.dqn import get_dqn_policy
from examples.cim.rl.config import action_shaping_conf, reward_shaping_conf, state_dim
from examples.cim.rl.env_sampler import CIMEnvSampler
learn_env, test_env = Env(scenario="cim", topology=topo, durations=dur), Env(scenario="cim", topology=topo, durations=dur)
n_ports = len(learn_env.agent_idx_list)
policies = [get_dqn_policy(state_dim, len(action_shaping_conf["action_space"]), f"dqn_{i}.policy") for i in range(n_ports)]
sampler = CIMEnvSampler(learn_env=learn_env, test_env=test_env, policies=policies,
                        agent2policy={agent: f"dqn_{agent}.policy" for agent in learn_env.agent_idx_list},
                        reward_eval_delay=reward_shaping_conf["time_window"])
sampler.sample(num_steps=50)
steps = exps = 0
t0 = time.perf_counter()
while time.perf_counter() - t0 < budget:
    res = sampler.sample(num_steps=200)
    steps += 200
    exps += sum(len(b) for b in res["experiences"])
print(steps, exps, time.perf_counter() - t0)
"""
    try:
        out = _run_reference(code, [topology, durations, budget_s], budget_s * 4 + 180)
        steps, exps, dt = out.stdout.split()[-3:]
        steps, exps, dt = int(steps), int(exps), float(dt)
    except Exception as e:
        return {"error": (str(e) + " " + (out.stderr[-300:] if "out" in dir() else ""))[:400]}
    return {"value": steps / dt, "unit": "env-steps/s", "cores": 1, "kind": "reference", "experiences_per_s": exps / dt,
            "sample": f"maro.rl CIMEnvSampler.sample(num_steps=200) x {steps // 200} on the reference Env of {topology} ({durations} ticks) with the example's "
                      f"{'per-port'} dueling DQN policies on PyTorch CPU (1 thread), single process: {steps} interactions, {exps} experiences in {dt:.1f} s",
            "measured": "live, in this run", "reference_from": reference_runtime()[3], **box_description()}


def bench_collect(args, engines, streams, qnet, chains, n, G, dev, rank, world, dist, reset_ms):
    """SURVEY.md 8(d) config 5 as the reference runs it: `AbsEnvSampler.sample(num_steps)` with on-device inference — here
    CimBatchSampler.sample_fused over every group's engine: per step ONE fused call for sampler state + per-port DQN + action
    translation (mrx_cim_dqn_act), the transition cache updated by one kernel (no host sync on the step path), then
    mrx_cim_step; per call the delayed rewards, per-agent next states, emission and episode roll-over.  One "step" of this
    bench = one interaction of every env of the rank (all groups).  Returns the JSON object on rank 0."""
    import torch

    from maro_amd.cim.sampler import CimBatchSampler, sample_fused_groups
    t_leg = time.perf_counter()
    samplers = []
    for g, e in enumerate(engines):
        with torch.cuda.stream(streams[g]):
            samplers.append(CimBatchSampler(e))
    seeds_of = [(lambda ep, g=g, e=e: ep * 1000003 + torch.arange(e.n_envs, dtype=torch.int64) + rank * n + g * 131071 + 1) for g, e in enumerate(engines)]

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def one_call(k):
        # the groups' step generators advanced in turn: every engine is bound to its group's stream (use_stream), the sampler puts
        # its own tensor ops on that stream too, and a step is a few C-ABI calls — the groups' kernels overlap like in the headline loop
        res = sample_fused_groups(samplers, qnet, k, seeds=seeds_of, reset_every=args.reset_every)
        return sum(int(r["tick"].shape[0]) for r in res)

    # untimed: into mid-episode (the first reward window must have passed for experiences to flow), then the warmup
    pre = 0
    while pre < max(args.preroll_ticks * 3, 256):
        one_call(128)
        pre += 128
    one_call(max(args.warmup, 1))
    vals, exps, dts = [], [], []
    for _ in range(max(1, args.repeats)):
        i0 = sum(int(s.interactions.item()) for s in samplers)
        sync_all()
        t0 = time.perf_counter()
        nexp = one_call(args.steps)
        sync_all()
        dt = time.perf_counter() - t0
        steps_done = sum(int(s.interactions.item()) for s in samplers) - i0
        t = torch.tensor([dt, float(steps_done), float(nexp)], dtype=torch.float64, device=dev)
        if dist is not None:
            tm = t[:1].clone()
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            t[0] = tm[0]
        dts.append(float(t[0])); vals.append(float(t[1]) / float(t[0])); exps.append(float(t[2]) / float(t[0]))
    # the policy launches alone (act of every group, back to back, no record / step): its own MFMA roofline
    b_act = [dict(a=torch.zeros((e.n_envs, 1, 4), dtype=torch.int32, device=dev), n=torch.zeros(e.n_envs, dtype=torch.int32, device=dev)) for e in engines]
    sync_all()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        for g in range(G):
            qnet[g].act(b_act[g]["a"], b_act[g]["n"])
    torch.cuda.synchronize(dev)
    act_ms = (time.perf_counter() - t0) / reps * 1e3
    deciding = sum(int((e.decisions[:, 7] == 1).sum().item()) for e in engines)
    # ---- N > 1: the learner-side collection (outside the timed windows): one more call, its experiences joined on rank 0
    # (gather_experiences_to_learner: ragged sizes, one grouped send / receive), and the return path: the learner's packed
    # networks broadcast to every rank's actors (broadcast_policy)
    exp_gather = policy_bcast = None
    if dist is not None:
        from maro_amd.cim.rollout import broadcast_policy, gather_experiences_to_learner
        sync_all()
        ta = time.perf_counter()
        res = sample_fused_groups(samplers, qnet, args.steps, seeds=seeds_of, reset_every=args.reset_every)
        sync_all()
        tb = time.perf_counter()
        mine = {k: torch.cat([r[k] for r in res]) for k in res[0] if k != "env_metric"}
        goff = [0] + [sum(e.n_envs for e in engines[:g + 1]) for g in range(G - 1)]
        mine["env_id"] = torch.cat([r["env_id"] + goff[g] for g, r in enumerate(res)])
        nbytes = sum(t.numel() * t.element_size() for t in mine.values())
        joined = gather_experiences_to_learner(mine, env_offset=rank * n, dst=0, transport=exchange_transport())
        sync_all()
        tc = time.perf_counter()
        tt = torch.tensor([tb - ta, tc - tb, float(nbytes), float(mine["tick"].shape[0])], dtype=torch.float64, device=dev)
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        if rank == 0:
            assert joined["tick"].shape[0] == int(tt[3].item())
            exp_gather = {"sample_ms": float(tmax[0]) * 1e3, "gather_ms": float(tmax[1]) * 1e3, "gather_share": float(tmax[1] / (tmax[0] + tmax[1])),
                          "bytes_all_ranks": float(tt[2]), "experiences": int(tt[3].item()), "steps_per_call": args.steps}
        del joined, mine, res
        sync_all()
        td = time.perf_counter()
        blob = broadcast_policy(qnet[0].weights.clone() if rank == 0 else None, qnet, src=0, transport=exchange_transport())
        sync_all()
        policy_bcast = {"ms": (time.perf_counter() - td) * 1e3, "bytes": blob.numel() * 4, "what": "rollout.broadcast_policy: one broadcast of the packed "
                        "networks + an in-place set_policy_state on every group's actor"}
    # ---- parity (untimed; rank 0's own envs): the loop's own elements of a few envs replayed on the C oracle
    parity = None
    if args.parity_envs > 0 and rank == 0:
        from tests.bench_parity import replay_collect_against_oracle
        parity = replay_collect_against_oracle(samplers, qnet, seeds_of, None, args.topology, k=min(args.parity_envs, 6), num_steps=384,
                                               reset_every=args.reset_every, chains=chains)
    if rank != 0:
        return None
    med = sorted(range(len(vals)), key=lambda i: vals[i])[len(vals) // 2]
    ms_per_step = dts[med] / args.steps * 1e3
    from maro_amd import _lib as mrx_lib
    code_key = f"{getattr(engines[0], 'code_object_key', None)}+{mrx_lib.source_hash('cim')}"   # step kernels' code object + the library's sources (DQN / sampler kernels)
    traffic, basis = measured_bytes_collect(n, G, code_key)
    achieved = None if traffic is None else traffic / (ms_per_step * 1e-3) / 1e9
    tf = DQN_FLOPS_PER_ENV * deciding / (act_ms * 1e-3) / 1e12
    out = {"metric": "env-steps/sec while collecting experiences (CimBatchSampler.sample_fused), CIM global_trade.22p",
           "value": vals[med], "unit": "env-steps/s", "experiences_per_s": exps[med], "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32+f64 (env), f32 MFMA (policy)",
           "data": "synthetic", "repeats": len(vals), "value_min": min(vals), "value_max": max(vals), "gpu_seconds_total": time.perf_counter() - t_leg,
           "config": {"workload": f"CIM {args.topology}, {n} envs/GPU x {world} GPU, durations {engines[0].durations}, maro.rl EnvSampler loop batched on device: per-port "
                                  f"dueling DQN (random-init, exact-f32 MFMA) + CIMEnvSampler state / reward shaping + transition cache + roll-over",
                      "envs_per_gpu": n, "groups_per_gpu": G, "reset_every": args.reset_every, "specialized_kernels": bool(engines[0].specialized), "code_object_key": code_key,
                      "reset_ms_whole_batch": reset_ms, "look_back": samplers[0].look_back, "reward_window": samplers[0].time_window,
                      "experience_gather": exp_gather, "policy_broadcast": policy_bcast,
                      "what_a_step_is": "one sample_fused interaction of every env: mrx_cim_dqn_act + the transition-cache update + mrx_cim_step; per call: mrx_cim_sampler_emit"},
           "roofline": {"bound": "hbm", "kernel": "the whole loop (mrx_k_cim_dqn_prep, mrx_k_cim_dqn_mlp16, mrx_k_cim_step_tab, mrx_k_cim_sampler_emit_all, ...)", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                        "unit": "GB/s", "frac": None if achieved is None else achieved / HBM_PEAK_GBPS, "traffic": traffic, "basis": basis,
                        "note": "bytes of one batch interaction (all kernels) / its wall time; null when no PMC record of this exact build exists"},
           "roofline_policy": {"bound": "mfma", "kernel": "mrx_k_cim_dqn_mlp16 (+ mrx_k_cim_dqn_prep: binning + state rows)", "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": tf / MFMA_F32_PEAK_TFLOPS, "dtype": "f32 (v_mfma_f32_16x16x4_f32)", "kernel_ms": act_ms, "deciding_envs": deciding,
                               "flops_per_env": DQN_FLOPS_PER_ENV, "basis": "algorithmic flops (2 x sum(in x out) of the example's layers x deciding envs) / wall time of the act "
                               "launches of all groups issued back to back with nothing else on the GPU"}}
    if parity is not None:
        out["parity"] = parity
    if world == 1 and not args.no_cpu:
        ref = cpu_baseline_collect(args.topology, engines[0].durations if engines[0].durations <= 1120 else 1120, min(args.cpu_seconds, 12.0))
        if ref is not None:
            out["cpu_baseline"] = ref
    return out


def _fastobj_built():
    try:
        from maro_amd import _fastobj  # noqa: F401
        return True
    except ImportError:
        return False


def bench_object_api(args, dev, n=16384, groups=3, steps=12, warmup=3, view_seconds=1.5):
    """What a MARO user gets from the reference-shaped OBJECT API: ``GpuVectorEnv(n, groups=3).step(list of
    Action) -> (metrics list, DecisionEvent list, done)`` — the counterpart of maro/vector_env/vector_env.py:116-144 — with the
    agent in Python reading every DecisionEvent and building one Action object per env, and ``env_view(0).step(Action)`` — what
    ``AbsEnvSampler(learn_env=view)`` calls (maro/rl/rollout/env_sampler.py:340-352).  env-steps/s of the whole loop (agent
    included), next to the tensor C-ABI's number in the headline.  Parity: sampled envs replayed on the C oracle with the very
    actions the agent took (every event, metric and done flag)."""
    import numpy as np
    import torch

    from maro_amd.cim.payloads import Action, ActionType
    from maro_amd.cim.vector_env import GpuVectorEnv
    topo, dur = args.topology, args.durations
    seeds = np.arange(n, dtype=np.int64) + 31
    env = GpuVectorEnv(n, "cim", topo, durations=dur, max_snapshots=args.ring, max_actions=1, seeds=seeds, groups=groups, device=dev, specialize=bool(args.specialize))
    LOAD, DISCHARGE = ActionType.LOAD, ActionType.DISCHARGE

    def agent(events, k):
        # a legal deterministic rule of the event alone: alternate a half load / half discharge by (step + port) parity
        out = []
        for ev in events:
            if ev is None:
                out.append(None)
            elif (k + ev.port_idx) & 1:
                out.append(Action(ev.vessel_idx, ev.port_idx, ev.action_scope.load >> 1, LOAD))
            else:
                out.append(Action(ev.vessel_idx, ev.port_idx, ev.action_scope.discharge >> 1, DISCHARGE))
        return out

    checked = sorted({0, 1, n // 2, n - 1})
    log = {e: [] for e in checked}
    metrics, events, done = env.step(None)
    t_agent = t_step = 0.0
    resolved = 0
    tuned = None
    for k in range(warmup + 2 * steps):
        if k == warmup + steps:
            # the same loop once more with the interpreter's cyclic collector out of the way of the agent's own object traffic
            # (gc.freeze() + a young-generation threshold above a step's allocations: two lines a user adds around a rollout)
            default_gc = (resolved, t_agent, t_step)
            from maro_amd.cim.vector_env import object_api_gc
            tuned = object_api_gc(n)
            tuned.__enter__()
        if k == warmup or k == warmup + steps:
            torch.cuda.synchronize(dev)
            t_agent = t_step = 0.0
            resolved = 0
        ta = time.perf_counter()
        actions = agent(events, k)
        tb = time.perf_counter()
        for e in checked:
            a, ev = actions[e], events[e]
            log[e].append(((ev.tick, ev.port_idx, ev.vessel_idx, ev.action_scope.load, ev.action_scope.discharge, ev.early_discharge), dict(metrics[e]),
                           None if a is None else (a.vessel_idx, a.port_idx, a.quantity, 0 if a.action_type is LOAD else 1)))
        tb2 = time.perf_counter()
        metrics, events, done = env.step(actions)
        tc = time.perf_counter()
        t_agent += tb - ta
        t_step += tc - tb2
        resolved += sum(1 for ev in events if ev is not None)
    # `value`: the loop as the API documents it — inside `with maro_amd.cim.vector_env.object_api_gc(n):` (gc.freeze() + a young-generation
    # threshold above one step's allocations); the same loop under CPython's default collector rides along
    tuned.__exit__()
    r0, a0, s0 = default_gc
    default_gc = {"value": r0 / (a0 + s0), "unit": "env-steps/s", "ms_agent_per_step": a0 / steps * 1e3, "ms_env_step_per_step": s0 / steps * 1e3,
                  "what": "the same loop with CPython's default collector thresholds: every 700th of the ~50 000 objects a step creates walks the young generation"}
    dt = t_agent + t_step        # (the parity log of the 4 sampled envs is not part of the loop)
    out = {"value": resolved / dt, "value_default_gc": default_gc, "object_builders": "C (maro_amd/_fastobj.so)" if _fastobj_built() else "Python comprehensions (maro_amd/_fastobj.so not built)", "unit": "env-steps/s", "envs": n, "groups": groups, "steps": steps, "ms_per_step": dt / steps * 1e3,
           "ms_agent_per_step": t_agent / steps * 1e3, "ms_env_step_per_step": t_step / steps * 1e3,
           "what": "GpuVectorEnv(n, groups).step(list of Action) inside `with object_api_gc(n):` — per step the Python agent reads every DecisionEvent (port, vessel, action_scope) "
                   "and builds one Action per env; step() encodes them, runs mrx_cim_step on every group, reads the results back once and builds the metrics dicts and DecisionEvents",
           "floor": "every env-step creates a DecisionEvent, a metrics dict and an Action and is touched by three Python loops (agent, encode, build): "
                    "~0.3-1 us of interpreter time per object, whatever the GPU does"}
    # parity: the sampled envs on the C oracle, driven by the agent's own actions
    from oracle.cim_oracle import CimOracle
    ok, first = True, None
    for e in checked:
        o = CimOracle(topo, durations=dur)
        o.set_seed(int(seeds[e]))
        o.reset(keep_seed=True)
        om, od, odone = o.step(None)
        for i, (evt, met, act) in enumerate(log[e]):
            want = tuple(int(x) for x in od[:6])
            if odone or want != evt or (int(om[0]), int(om[1]), int(om[2])) != (met["order_requirements"], met["container_shortage"], met["operation_number"]):
                ok, first = False, first or {"env": e, "step": i, "oracle": want, "object_api": evt}
                break
            om, od, odone = o.step([act] if act is not None else None)
    out["parity"] = {"ok": ok, "envs_checked": len(checked), "env_steps_checked": sum(len(v) for v in log.values()), "first_mismatch": first,
                     "what": "every DecisionEvent (tick, port, vessel, action scope, early discharge) and metrics dict the agent saw for the sampled envs vs the C oracle driven by the agent's own actions"}
    # ---- one env through its view: what AbsEnvSampler(learn_env=env_view(0)) does per interaction
    one = GpuVectorEnv(1, "cim", topo, durations=dur, max_snapshots=args.ring, max_actions=1, seeds=[7], device=dev, specialize=bool(args.specialize))
    view = one.env_view(0)
    m, ev, d = view.step(None)
    k = steps_v = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < view_seconds:
        if d:
            view.reset()
            m, ev, d = view.step(None)
            continue
        a = Action(ev.vessel_idx, ev.port_idx, ev.action_scope.load >> 1, LOAD) if k & 1 else Action(ev.vessel_idx, ev.port_idx, ev.action_scope.discharge >> 1, DISCHARGE)
        _ = view.tick, view.frame_index
        m, ev, d = view.step(a)
        k += 1
        steps_v += 1
    dtv = time.perf_counter() - t0
    out["env_view"] = {"value": steps_v / dtv, "unit": "env-steps/s", "envs": 1, "us_per_step": dtv / max(steps_v, 1) * 1e6,
                       "what": "GpuVectorEnv(1).env_view(0).step(Action) + tick + frame_index per interaction: one launch and one read-back per step (latency-bound: a single env cannot fill a GPU)"}
    del env, one
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenario", default="cim", choices=["cim", "citi_bike"])
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs", type=int, default=None, help="environments per GPU (weak scaling); default 16384 (cim) / 4096 (citi_bike)")
    ap.add_argument("--policy", default="random", choices=["random", "dqn"],
                    help="random: the device random legal agent (headline); dqn: CIMEnvSampler state + 22 per-port dueling DQNs "
                         "+ action translation, all on the device (SURVEY.md 8d config 5; use --ring 8 or more)")
    ap.add_argument("--collect", action="store_true", help="with --policy dqn: time the EXPERIENCE-COLLECTING loop (CimBatchSampler.sample_fused: "
                    "maro.rl's AbsEnvSampler.sample batched — transition cache, per-agent next states, delayed rewards, episode roll-over) "
                    "instead of the bare act -> step loop; reports experiences/s next to env-steps/s")
    ap.add_argument("--reset-every", type=int, default=32, help="--collect: envs whose episode ended are finalised / reset every this many steps")
    ap.add_argument("--obs", default="fused", choices=["fused", "query"],
                    help="how the per-step ports / deciding-vessel snapshot slices are produced: fused into the step kernel, or by mrx_cim_query")
    ap.add_argument("--agent", default="launch", choices=["fused", "launch"], help="cim, random agent: a mrx_cim_random_policy launch before every step (default), or answered inside "
                    "the step kernel (mrx_cim_set_device_agent: one launch per batch step — measured 3-5 %% slower, profiles/r06_experiments.md)")
    ap.add_argument("--graphs", type=int, default=0, help="1: capture one step per group in a hipGraph and replay it (cim)")
    ap.add_argument("--groups", type=int, default=3, help="independent env groups per GPU, each on its own HIP stream (cim)")
    ap.add_argument("--step-mode", type=int, default=0, help="launch form of mrx_cim_step (mrx_cim_set_step_mode): 0 default (sorted), "
                    "1 unsorted, 2 sorted, 4 split")
    ap.add_argument("--topology", default="global_trade.22p_l0.8")
    ap.add_argument("--replay-period", type=int, default=1, help="citi_bike wave kernels: the replay kernel on every n-th batch step (mrx_cb_set_replay_period)")
    ap.add_argument("--replay-overlap", type=int, default=0, help="citi_bike wave kernels: replay kernel beside the in-tick kernel (mrx_cb_set_replay_overlap; 0 = after it)")
    ap.add_argument("--step-budget", type=int, default=0, help="citi_bike: bounded steps for the main window (mrx_cb_set_step_budget; 0 = every call yields a decision)")
    ap.add_argument("--bounded-budget", type=int, default=24, help="citi_bike: budget of the extra bounded-steps leg (0: skip it)")
    ap.add_argument("--cb-groups", type=int, default=1, help="citi_bike: independent env groups per GPU, each engine on its own HIP stream")
    ap.add_argument("--durations", type=int, default=1120)
    ap.add_argument("--preroll-ticks", type=int, default=300, help="untimed steps before the warmup until the batch's mean tick reaches this "
                    "(the timed window then measures mid-episode steady state, whatever --steps / --warmup are)")
    ap.add_argument("--specialize", type=int, default=1, help="1: step with kernels compiled for this exact plan (maro_amd/cim/specialize.py; "
                    "built by __graft_entry__.build() for the default workload, else ~3 s of hipcc at engine creation); 0: generic kernels")
    ap.add_argument("--ring", type=int, default=4, help="snapshot ring slots per env (max_snapshots)")
    ap.add_argument("--no-query", action="store_true", help="skip the per-step snapshot slice")
    ap.add_argument("--no-episode", action="store_true", help="skip the end-to-end leg (reset + one full episode of the whole batch)")
    ap.add_argument("--sustained-episodes", type=int, default=6, help="sustained leg (after the end-to-end leg): whole episodes per group run back to back, "
                    "resets overlapped with the other groups' steps (0: skip)")
    ap.add_argument("--episode-block", type=int, default=32, help="(kept for old command lines; the end-to-end leg polls every group every 32 batch steps without a device-wide sync)")
    ap.add_argument("--parity-envs", type=int, default=64, help="envs replayed on the CPU oracle after the run (0: off)")
    ap.add_argument("--repeats", type=int, default=5, help="the timed window (exactly --steps steps between barrier + synchronize) is run this many "
                    "times back to back; `value` is the median window, value_min / value_max the spread")
    ap.add_argument("--gather-every", type=int, default=0, help="N > 1 ranks: additionally roll out K-step trajectories and gather each to the learner "
                    "(gather_to_learner), reporting the gather's share of a rollout (0: only the single 32-step gather)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--object-api", action="store_true", help="run ONLY the object-API leg (GpuVectorEnv.step(list of Action), env_view stepping) and print it")
    ap.add_argument("--secondary", type=int, default=-1, help="after the headline (cim, random agent), also run BASELINE configs 4 and 5 as short legs and embed "
                    "them under `secondary` in the same JSON line (citi_bike toy.3s_4t 4096 envs — and, on one GPU, all 32768 of config 4; DQN collection loop 8192 envs), each with parity, "
                    "cpu_baseline and a measured-bytes roofline.  -1 = auto: on for the default workload (what the driver runs), off when a flag selects another one")
    args = ap.parse_args()
    explicit_envs = args.envs is not None
    if args.envs is None:
        args.envs = 16384 if args.scenario == "cim" else 4096
    if args.secondary < 0:
        args.secondary = int(args.scenario == "cim" and args.policy == "random" and not args.collect and not explicit_envs and not args.graphs
                             and args.topology == "global_trade.22p_l0.8" and args.obs == "fused" and not args.no_query)

    import torch

    import __graft_entry__ as ge
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist, dev = init_dist(world, local_rank, build=ge.build)
    torch.cuda.set_device(dev)
    if args.object_api:
        ge.build()
        out = bench_object_api(args, dev, n=args.envs, groups=args.groups)
    elif args.scenario == "citi_bike":
        out = bench_citi_bike(args, dist, dev, rank, world)
    else:
        out = bench_cim(args, dist, dev, rank, world)
        if args.secondary:   # (every rank takes part: the legs hold collectives)
            import copy
            import gc
            sec = {}
            gc.collect()
            torch.cuda.empty_cache()
            # BASELINE.json configs[3]: citi_bike toy.3s_4t, 4096 envs per GPU (32768 over 8 GPUs)
            a4 = copy.copy(args)
            a4.scenario, a4.envs, a4.topology, a4.durations, a4.step_budget, a4.bounded_budget = "citi_bike", 4096, "toy.3s_4t", 1120, 0, 0
            a4.parity_envs = min(args.parity_envs, 8)
            r4 = bench_citi_bike(a4, dist, dev, rank, world)
            gc.collect()
            torch.cuda.empty_cache()
            # the WHOLE of configs[3] (32768 envs) on one GPU: it fits, and 4096 envs put 512 one-wave workgroups on 1024 SIMDs
            r4w = None
            if world == 1:
                a4w = copy.copy(a4)
                a4w.envs, a4w.no_cpu = 32768, True
                r4w = bench_citi_bike(a4w, dist, dev, rank, world)
                gc.collect()
                torch.cuda.empty_cache()
            # citi_bike at the size the reference SHIPS (ny.*: 800 stations, filter chain 80 -> 40 -> 20): city.800s, sustained — a window
            # that spans several decision ticks, bounded steps, the wave-cooperative kernels (the toy of configs[3] never leaves the lane kernel)
            r8 = None
            if world == 1:
                a8 = copy.copy(a4)
                a8.topology, a8.envs, a8.durations, a8.steps, a8.warmup, a8.repeats = "city.800s", 4096, 2880, 900, 300, 3   # (value = the median window, as in profiles/*_citi_bike.md)
                a8.step_budget, a8.replay_period, a8.cb_groups, a8.bounded_budget, a8.specialize, a8.no_cpu = 96, 4, 2, 0, 1, True   # (the replay kernel on every 4th batch step, four steps' worth of records per call; two env groups on their own streams, their replay calls staggered)
                try:
                    r8 = bench_citi_bike(a8, dist, dev, rank, world)
                except Exception as e:      # (a plan this size compiles for minutes when the in-tree cache misses: never a reason to fail the bench)
                    r8 = {"error": repr(e)[:300]}
                gc.collect()
                torch.cuda.empty_cache()
            # BASELINE.json configs[4]: CIM 22p + the maro.rl DQN EnvSampler loop, 8192 envs per GPU (65536 over 8 GPUs), on-device inference
            a5 = copy.copy(args)
            a5.policy, a5.collect, a5.envs, a5.ring, a5.no_episode = "dqn", True, 8192, max(args.ring, 8), True
            a5.groups = 2    # (a latency-bound three-launch chain per group: measured 1 / 2 / 3 groups = 59 / 65 / 62 M at this size)
            a5.parity_envs = min(args.parity_envs, 6)
            r5 = bench_cim(a5, dist, dev, rank, world)
            r_obj = None
            if world == 1:
                gc.collect()
                torch.cuda.empty_cache()
                try:
                    r_obj = bench_object_api(args, dev)
                except Exception as e:      # a convenience leg, never a reason to fail the bench
                    r_obj = {"error": repr(e)[:300]}
            if out is not None:
                sec["citi_bike_config4"], sec["collect_config5"] = r4, r5
                if r_obj is not None:
                    if "value" in r_obj and "cpu_baseline_reference" in out and isinstance(out["cpu_baseline_reference"].get("vector_env"), dict):
                        r_obj["cpu_baseline"] = dict(out["cpu_baseline_reference"]["vector_env"], kind="reference", note="maro.vector_env.VectorEnv on this box's cores (headline's cpu_baseline_reference.vector_env)")
                    sec["object_api"] = r_obj
                if r8 is not None:
                    if "config" in r8:
                        r8["config"]["what"] = "citi_bike at the reference's own topology size (800 stations, ny filter chain), sustained over several decision ticks; cpu_baseline: see citi_bike_config4"
                    sec["citi_bike_city800"] = r8
                if r4w is not None:
                    r4w["config"]["what"] = "BASELINE.json configs[3] whole (32768 envs) on ONE GPU; cpu_baseline: see citi_bike_config4"
                    sec["citi_bike_config4_one_gpu"] = r4w
                out["secondary"] = sec
                out["gpu_seconds_total"] += sum((r or {}).get("gpu_seconds_total", 0.0) for r in (r4, r4w, r8, r5))
    if rank == 0 and out is not None:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def bench_cim(args, dist, dev, rank, world):
    """The CIM legs: the headline (random legal agent), `--policy dqn` (act -> step) and `--policy dqn --collect` (the whole
    experience-collecting loop).  Returns the JSON object on rank 0, None on the other ranks."""
    import torch

    # The per-GPU batch is split into G independent groups, each with its own engine and HIP stream: the kernels of the
    # other groups fill a group's launch gaps and tail.  Envs never interact, so this is pure scheduling.
    n, G = args.envs, max(1, args.groups)
    # the episode must outlast the run (finished envs would idle): ~0.45 ticks per env-step on 22p
    need_ticks = args.preroll_ticks + int(0.55 * (args.warmup + args.steps * max(1, args.repeats) + min(args.steps, 100) + 64)) + 64
    sim_durations = max(args.durations, need_ticks)
    obs_mode = "none" if args.no_query else args.obs
    # --agent fused: the random legal agent answered inside the step kernel (one launch per batch step and group); no policy launch
    # carries the sorted launch's order list then, so the step takes the unsorted form
    agent = "fused" if (args.agent == "fused" and args.policy == "random" and not args.graphs) else "launch"
    step_mode = 1 if (agent == "fused" and args.step_mode == 0) else args.step_mode
    engines, streams, bufs, sizes, offs = build_cim_groups(args.topology, n, G, dev, rank, sim_durations, args.ring, args.specialize, step_mode,
                                                           obs_mode, args.policy, agent)
    topo = engines[0].topo
    ports = torch.arange(topo.n_ports, dtype=torch.int32, device=dev)
    qnet = None
    if args.policy == "dqn":
        # the CIM RL example's rollout path entirely on the device — CIMEnvSampler state
        # (look-back snapshot slices), 22 per-port dueling DQNs (random-init weights, exact-f32 MFMA), greedy action and the
        # env_sampler.py action translation, fused in mrx_cim_dqn_act (maro_amd/csrc/cim_dqn.h)
        from maro_amd.cim.policy import ACTION_SPACE, FusedPerPortDQN, random_chains
        from maro_amd.cim.sampler import CimBatchSampler
        chains = random_chains(topo.n_ports, CimBatchSampler(engines[0]).state_dim, len(ACTION_SPACE), seed=0)
        qnet = [FusedPerPortDQN(e, chains) for e in engines]
    torch.cuda.synchronize(dev)

    def group_seeds(g):
        return torch.arange(sizes[g], dtype=torch.int64) + rank * n + offs[g] + 1

    def reset_all():
        """Env.reset for the whole batch: route unrolling, order proportion and — with the order table — every order of the
        episode are generated on the device.  Returns the wall time in ms."""
        torch.cuda.synchronize(dev)
        t_r = time.perf_counter()
        for g, eng in enumerate(engines):
            eng.reset(group_seeds(g))
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t_r) * 1e3

    reset_ms = reset_all()
    graphs = [None] * G
    if args.collect:
        if qnet is None:
            raise SystemExit("--collect needs --policy dqn")
        return bench_collect(args, engines, streams, qnet, chains, n, G, dev, rank, world, dist, reset_ms)

    def one_step(i, g, timing=None, count=True):
        eng, b, st = engines[g], bufs[g], streams[g]
        if i == 0:
            if agent == "fused":   # (the agent's key restarts with the episode: the first answered decision is keyed 1)
                eng.set_device_agent(b["actions"], b["n_actions"], b["counts"], next_key=1)
            eng.step()  # first step of the episode: action=None
            return
        if agent == "fused":
            if timing is not None:
                timing[2].record(st)
                timing[0].record(st)
            eng.step(b["actions"], b["n_actions"])   # answers the decisions it raises: the next step's actions are in place
            if timing is not None:
                timing[1].record(st)
            return
        if graphs[g] is not None and timing is None:
            with torch.cuda.stream(st):
                graphs[g].replay()  # policy -> step -> snapshot slices, captured once (hipGraph)
            return
        ctr = b["counter"] if (timing is None and count) else None
        if qnet is not None:
            # mrx_cim_dqn_act: state gather + MFMA MLP + argmax + translation
            if timing is not None:
                timing[2].record(st)
            qnet[g].act(b["actions"], b["n_actions"], counter=ctr)
        else:
            eng.random_policy(-1 if args.graphs else i, b["actions"], b["n_actions"], ctr)
        if timing is not None:
            timing[0].record(st)
        eng.step(b["actions"], b["n_actions"])
        if timing is not None:
            timing[1].record(st)
        if b["q_ports"] is not None:
            # ports[frame::7 attrs] and vessels[frame:vessel:3 attrs] of the pending decision
            eng.query("ports", eng.decisions[:, 6:7], ports, QUERY_ATTRS, out=b["q_ports"])
            eng.query("vessels", eng.decisions[:, 6:7], eng.decisions[:, 2:3], VESSEL_QUERY_ATTRS, out=b["q_vessel"])

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def total_ticks():
        return sum(e.ticks.to(torch.int64).sum().item() for e in engines)

    # ---- untimed: into mid-episode (vessels loaded, discharge records and return rings populated), then the warmup
    step_i = 0
    torch.cuda.synchronize(dev)
    t_pre = time.perf_counter()
    while True:
        for _ in range(64):
            for g in range(G):
                one_step(step_i, g)
            step_i += 1
        torch.cuda.synchronize(dev)
        if total_ticks() / n >= args.preroll_ticks or step_i > 8 * args.preroll_ticks + 64:
            break
    preroll_steps = step_i
    for _ in range(args.warmup):
        for g in range(G):
            one_step(step_i, g)
        step_i += 1
    sync_all()
    if args.graphs:
        # the launch-bound inner loop (policy + step (+ slices) per group and step) is captured once per group and replayed
        for g in range(G):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=streams[g]):
                eng, b = engines[g], bufs[g]
                if qnet is not None:
                    qnet[g].act(b["actions"], b["n_actions"], counter=b["counter"])
                else:
                    eng.random_policy(-1, b["actions"], b["n_actions"], b["counter"])
                eng.step(b["actions"], b["n_actions"])
                if b["q_ports"] is not None:
                    eng.query("ports", eng.decisions[:, 6:7], ports, QUERY_ATTRS, out=b["q_ports"])
                    eng.query("vessels", eng.decisions[:, 6:7], eng.decisions[:, 2:3], VESSEL_QUERY_ATTRS, out=b["q_vessel"])
            graphs[g] = gr
        sync_all()
        for _ in range(10):
            for g in range(G):
                one_step(step_i, g)
            step_i += 1
        sync_all()
    # ---- the timed window: exactly --steps steps between barrier + synchronize on both sides; run --repeats times back to
    # back (value = the median window), so that a driver run with a handful of steps is not a single 1-2 ms sample
    gpu_busy_s = {"preroll_and_warmup": time.perf_counter() - t_pre, "timed_windows": 0.0}   # host clock around back-to-back GPU work
    windows = []
    for rep in range(max(1, args.repeats)):
        for b in bufs:
            b["counter"].zero_()
        tick0 = total_ticks()
        if rep == 0:
            tick_mean0 = tick0 / n
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            for g in range(G):
                one_step(step_i, g)
            step_i += 1
        t_issued = time.perf_counter() - t0   # host time to enqueue every launch of the window (the loop never waits for the GPU)
        sync_all()
        dt = time.perf_counter() - t0
        gpu_busy_s["timed_windows"] += dt
        # decisions answered inside the window (K policy calls per group in the window)
        windows.append({"dt": dt, "resolved": sum(int(b["counter"].item()) for b in bufs), "ticks": total_ticks() - tick0, "t_issued": t_issued})
    n_done = sum(int(e.done.sum().item()) for e in engines)
    status_bad = sum(int((e.status != 0).sum().item()) for e in engines)

    # ---- dominant kernel (the step kernel) timed live with HIP events, each pair on the stream of its launch, in the same
    # interleaved schedule as the timed loop (informational: roofline.achieved comes from the timed window above)
    reps = min(args.steps, 100)
    ev = [[tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(G)] for _ in range(reps)]
    sync_all()
    t_ev = time.perf_counter()
    for r in range(reps):
        for g in range(G):
            one_step(step_i, g, timing=ev[r][g])
        step_i += 1
    torch.cuda.synchronize(dev)
    gpu_busy_s["event_timed_steps"] = time.perf_counter() - t_ev
    durs = [a.elapsed_time(b) for row in ev for a, b, _ in row]
    policy_ms = sum(c.elapsed_time(a) for row in ev for a, _, c in row) / len(durs) if qnet is not None else None
    step_kernel_ms = sum(durs) / len(durs)                      # mean duration of one launch (ng envs; sorted launch: + the schedule kernel)
    span_ms = max([ev[0][g][0].elapsed_time(ev[-1][g2][1]) for g in range(G) for g2 in range(G)]) if G > 1 else sum(durs)
    in_flight = max(1.0, sum(durs) / span_ms) if G > 1 else 1.0    # mean number of step kernels running concurrently

    # ---- the one exchange step of a sharded rollout (north_star: RCCL "only to gather trajectories to the learner"): 32 steps of
    # (decision, action, metrics, done, fused observation) per env from every rank to rank 0 in ONE collective
    # (maro_amd/cim/rollout.py::gather_to_learner); outside the timed env-step window
    gather_ms = gather_bytes = gather_every = None
    if dist is not None:
        from maro_amd.cim.rollout import gather_to_learner

        def alloc_traj(T):
            traj = {"decisions": torch.zeros((T, n, 8), dtype=torch.int32, device=dev), "actions": torch.zeros((T, n, 1, 4), dtype=torch.int32, device=dev),
                    "metrics": torch.zeros((T, n, 3), dtype=torch.int64, device=dev), "done": torch.zeros((T, n), dtype=torch.uint8, device=dev)}
            if bufs[0].get("obs") is not None:
                traj["obs_ports"] = torch.zeros((T, n, topo.n_ports * len(QUERY_ATTRS)), dtype=torch.float32, device=dev)
            return traj

        def record_rollout(traj):
            nonlocal step_i
            for k in range(traj["done"].shape[0]):
                for g in range(G):
                    one_step(step_i, g, count=False)
                    with torch.cuda.stream(streams[g]):
                        sl = slice(offs[g], offs[g] + sizes[g])
                        traj["decisions"][k, sl], traj["actions"][k, sl] = engines[g].decisions, bufs[g]["actions"]
                        traj["metrics"][k, sl], traj["done"][k, sl] = engines[g].metrics, engines[g].done
                        if "obs_ports" in traj:
                            traj["obs_ports"][k, sl] = bufs[g]["obs"][0].reshape(sizes[g], -1)
                step_i += 1

        traj = alloc_traj(32)
        record_rollout(traj)
        sync_all()
        tg = time.perf_counter()
        gathered = gather_to_learner(traj, dst=0, sizes=[n] * world, transport=exchange_transport())   # weak scaling: every rank owns n envs, no size exchange
        torch.cuda.synchronize(dev)
        gather_ms = (time.perf_counter() - tg) * 1e3
        gather_bytes = sum(t.numel() * t.element_size() for t in traj.values())
        if rank == 0:
            assert gathered["decisions"].shape[1] == n * world
        del gathered, traj
        if args.gather_every > 0:
            # the learner's view of a sharded rollout loop: K steps of stepping (+ recording), then ONE gather, repeated;
            # max over ranks of each part, so the first real 8-GPU run reports the gather's share of a rollout, not one gather
            K = args.gather_every
            traj = alloc_traj(K)
            t_roll = t_gath = 0.0
            rounds = 4
            for _ in range(rounds):
                sync_all()
                ta = time.perf_counter()
                record_rollout(traj)
                sync_all()
                tb = time.perf_counter()
                gathered = gather_to_learner(traj, dst=0, sizes=[n] * world, transport=exchange_transport())
                sync_all()
                tc = time.perf_counter()
                t_roll += tb - ta
                t_gath += tc - tb
                del gathered
            tt = torch.tensor([t_roll, t_gath], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_roll, t_gath = (float(x) / rounds for x in tt.tolist())
            gather_every = {"steps_per_rollout": K, "rollout_ms": t_roll * 1e3, "gather_ms": t_gath * 1e3, "gather_share": t_gath / (t_roll + t_gath),
                            "bytes_per_rank": sum(t.numel() * t.element_size() for t in traj.values()), "rounds": rounds}
            del traj

    # ---- end to end (untimed window of its own): reset of the whole batch + one complete episode of every env, the same
    # agent; envs that finish early keep being stepped (they report done) until the slowest one is through
    episode = None
    if not args.no_episode and qnet is None and not args.graphs:
        if sim_durations != args.durations:   # (only very long runs stretch the episode: measure the nominal one on fresh engines)
            del engines[:]
            torch.cuda.empty_cache()
            ep_engines, streams, bufs, sizes, offs = build_cim_groups(args.topology, n, G, dev, rank, args.durations, args.ring, args.specialize,
                                                                      step_mode, obs_mode, args.policy, agent)
            engines.extend(ep_engines)
        for b in bufs:
            b["counter"].zero_()
        sync_all()
        t_e = time.perf_counter()
        # each group resets on ITS OWN stream and starts stepping as soon as its own reset (reset kernel + order table) is done:
        # no batch-wide barrier between reset and the first step, so one group's order-table generation (VALU-bound, 7 KB of LDS
        # per env) runs under the other groups' step kernels (latency-bound).  The wall time of a synchronised whole-batch reset
        # is reported separately (config.reset_ms_whole_batch, measured at the start of the run).
        for g, eng in enumerate(engines):
            eng.reset(group_seeds(g))
        ep_reset_ms = None
        # no device-wide synchronisation inside the episode: every group steps in blocks of 32 and polls its own done flag through a
        # pinned word and an event (the flag of block j is read after block j + 1 was enqueued); a group stops as soon as its envs are
        # through, the leg ends when every group has
        blk, k = 32, 0
        e_flags = [torch.zeros(2, dtype=torch.uint8).pin_memory() for _ in range(G)]
        e_evs = [[torch.cuda.Event(), torch.cuda.Event()] for _ in range(G)]
        e_pending, e_done, e_k = [None] * G, [False] * G, [0] * G
        slot = 0
        while not all(e_done) and k <= 8 * args.durations:
            for g in range(G):
                if e_done[g]:
                    continue
                for _ in range(blk):
                    one_step(e_k[g], g)
                    e_k[g] += 1
                if e_pending[g] is not None:
                    e_evs[g][e_pending[g]].synchronize()
                    if e_flags[g][e_pending[g]].item():
                        e_done[g] = True
                        continue
                with torch.cuda.stream(streams[g]):
                    e_flags[g][slot].copy_(engines[g].done.min().to(torch.uint8), non_blocking=True)
                    e_evs[g][slot].record(streams[g])
                e_pending[g] = slot
            slot ^= 1
            k += blk
        k = max(e_k)
        sync_all()
        t_ep = time.perf_counter() - t_e
        ep_steps = sum(int(b["counter"].item()) for b in bufs)
        episode = {"env_steps": ep_steps, "seconds": t_ep, "reset_ms": ep_reset_ms, "batch_steps": k}
        gpu_busy_s["end_to_end_episode"] = t_ep

    # ---- sustained: >= 1 s of whole episodes back to back, resets included and OVERLAPPED — what a rollout user gets.  Each group
    # runs episode after episode on its own stream and is reset (next seeds) as soon as a polled flag says its envs are all done;
    # the groups are staggered by a third of an episode (untimed pre-roll), so one group's reset kernels (order table: VALU-bound)
    # run under the other groups' step kernels (latency-bound).  No device-wide synchronisation inside the window: the done flag
    # of block k is read when block k + 1 has been enqueued.
    sustained = None
    if episode is not None and args.sustained_episodes > 0:
        nominal, blk = episode["batch_steps"], 32
        per_group_steps = args.sustained_episodes * nominal
        sync_all()
        for g, eng in enumerate(engines):
            eng.reset(group_seeds(g))
        kk = [0] * G                      # step index within the group's current episode
        ep_n = [0] * G                    # episodes the group has started after the first
        for g in range(1, G):
            for _ in range(g * nominal // G):
                one_step(kk[g], g, count=False)
                kk[g] += 1
        flags = [torch.zeros(2, dtype=torch.uint8).pin_memory() for _ in range(G)]
        evs = [[torch.cuda.Event(), torch.cuda.Event()] for _ in range(G)]
        pending = [None] * G              # which of the group's two flag slots holds an unread poll
        sync_all()
        for b in bufs:
            b["counter"].zero_()
        sync_all()
        t_s = time.perf_counter()
        done_steps = [0] * G
        resets_in_window = 0
        slot = 0
        while min(done_steps) < per_group_steps:
            for g in range(G):
                if done_steps[g] >= per_group_steps:
                    continue
                for _ in range(blk):
                    one_step(kk[g], g)
                    kk[g] += 1
                done_steps[g] += blk
                # read the PREVIOUS poll of this group (its event has long fired: a whole block of every group was enqueued since)
                if pending[g] is not None:
                    evs[g][pending[g]].synchronize()
                    if flags[g][pending[g]].item():
                        ep_n[g] += 1
                        resets_in_window += 1
                        engines[g].reset(group_seeds(g) + 1000003 * ep_n[g])
                        kk[g] = 0
                        pending[g] = None
                        continue
                with torch.cuda.stream(streams[g]):
                    flags[g][slot].copy_(engines[g].done.min().to(torch.uint8), non_blocking=True)
                    evs[g][slot].record(streams[g])
                pending[g] = slot
            slot ^= 1
        sync_all()
        t_sus = time.perf_counter() - t_s
        sus_steps = sum(int(b["counter"].item()) for b in bufs)
        sustained = {"env_steps": sus_steps, "seconds": t_sus, "resets": resets_in_window, "batch_steps_per_group": per_group_steps, "poll_every": blk}
        gpu_busy_s["sustained_episodes"] = t_sus

    # ---- parity (untimed): a sample of envs of THIS configuration replayed on the CPU oracle — every decision, metric,
    # fused observation and the final snapshot ring (tests/bench_parity.py; the oracle is the checker, never the thing measured)
    parity = None
    if args.parity_envs > 0 and qnet is None and not args.graphs and world == 1:
        from tests.bench_parity import replay_against_oracle
        parity = replay_against_oracle(engines, bufs, streams, sizes, offs, rank * n, args.topology, engines[0].durations, args.parity_envs,
                                       obs=bufs[0].get("obs") is not None)

    R = len(windows)
    t_max = torch.tensor([w["dt"] for w in windows], dtype=torch.float64, device=dev)
    ep_t = torch.tensor([episode["seconds"] if episode else 0.0], dtype=torch.float64, device=dev)
    tot = torch.tensor([float(w["resolved"]) for w in windows] + [float(w["ticks"]) for w in windows] +
                       [float(n_done), float(status_bad), float(episode["env_steps"] if episode else 0)], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        dist.all_reduce(ep_t, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    # every rank's own contribution to every window (the line's value must be sum(env-steps) / max(seconds) over the ranks)
    mine = torch.tensor([[w["dt"], float(w["resolved"])] for w in windows], dtype=torch.float64, device=dev)
    per_rank = [mine.clone() for _ in range(world)] if dist is not None else [mine]
    if dist is not None:
        dist.all_gather(per_rank, mine)
    dts = [float(x) for x in t_max.tolist()]
    tl = [float(x) for x in tot.tolist()]
    res_w, tick_w = tl[:R], tl[R:2 * R]
    n_done, status_bad, ep_steps_all = tl[2 * R:]
    vals = [res_w[i] / dts[i] for i in range(R)]
    med = sorted(range(R), key=lambda i: vals[i])[R // 2]      # the median window (whole-job rate: max-over-ranks time, summed counts)
    dt, resolved, ticks_adv, t_issued = dts[med], res_w[med], tick_w[med], windows[med]["t_issued"]

    if rank == 0:
        value = resolved / dt
        ms_per_step = dt / args.steps * 1e3
        tbar = ticks_adv / max(resolved, 1.0)  # mean ticks advanced per env-step
        F = frame_bytes(topo)
        b_step = (3.0 + tbar) * F + 4.0 * topo.n_targets * tbar + 40.0  # fixed per-env-step formula
        ng = n / G                                  # env-steps per launch
        kernel = ("mrx_k_cim_step_tab" if engines[0].layout.order_table_on else "mrx_k_cim_step") + ("_obs" if bufs[0].get("obs") is not None else "")
        # HBM bytes one launch really moves: the committed PMC passes of this same workload AND this same code object
        # (tools/gpu_profile.sh -> tools/refresh_pmc.py -> profiles/latest_pmc.json, stamped with the step kernels' cache key =
        # hash of plan text + device sources + flags + toolchain); any other build falls back to the algorithmic bytes
        traffic = pmc_src = None
        pmc_state = "no profiles/latest_pmc.json"
        code_key = getattr(engines[0], "code_object_key", None)
        try:
            with open(os.path.join(REPO, "profiles", "latest_pmc.json")) as fp:
                pmc = json.load(fp)
            if not (pmc["topology"] == args.topology and abs(pmc["envs_per_launch"] - ng) <= 1 and pmc.get("kernel") == kernel):
                pmc_state = "the record is of another configuration"
            elif pmc.get("code_object_key") != code_key or code_key is None:
                pmc_state = f"the record is of another build (code object {pmc.get('code_object_key')}, running {code_key})"
            else:
                traffic = (2.0 * pmc["fetch_size_kib"] + pmc["write_size_kib"]) * 1024.0
                pmc_src = {k: pmc.get(k) for k in ("source", "git_head", "code_object_key", "code_object_sha16", "bench_value", "date")}
                pmc_state = "matched"
        except Exception:
            pass
        # the measured ceiling of this access pattern (tools/hbm_pattern_bench: same grid, LDS reservation, pieces and streams, no
        # simulation work), committed by the same profile run
        ceiling = None
        try:
            with open(os.path.join(REPO, "profiles", "pattern_ceiling.json")) as fp:
                pc = json.load(fp)
            if pc.get("topology") == args.topology and abs(pc["envs_per_launch"] - ng) <= 1:
                ceiling = pc
        except Exception:
            pass
        alg_per_launch = b_step * (resolved / world / args.steps / G)   # SURVEY formula bytes of one launch (its env-steps x B_step)
        alg_gbps = alg_per_launch * G / (ms_per_step * 1e-3) / 1e9
        if traffic is not None:
            # achieved = bytes the kernel really moves per launch x launches per step / the TIMED window's ms_per_step (per GPU)
            achieved = traffic * G / (ms_per_step * 1e-3) / 1e9
            basis = "measured HBM bytes (PMC: 2 x FETCH_SIZE + WRITE_SIZE per launch) x launches per step / ms_per_step of the timed window"
        else:
            # the record of the SAME code object at another batch size: its bytes per env-step carry over (the per-env access list
            # does not depend on the batch); any other build has no measured bytes -> no fraction (never the SURVEY formula)
            achieved = basis = None
            try:
                if pmc.get("code_object_key") == code_key and code_key is not None and pmc.get("kernel") == kernel and pmc["topology"] == args.topology:
                    traffic = (2.0 * pmc["fetch_size_kib"] + pmc["write_size_kib"]) * 1024.0 / pmc["envs_per_launch"] * ng
                    achieved = traffic * G / (ms_per_step * 1e-3) / 1e9
                    pmc_src = {k: pmc.get(k) for k in ("source", "git_head", "code_object_key", "code_object_sha16", "bench_value", "date")}
                    basis = (f"measured HBM bytes per env-step of the same code object (PMC record at {pmc['envs_per_launch']:.0f} envs per launch) scaled to "
                             f"{ng:.0f} envs per launch x launches per step / ms_per_step of the timed window")
            except Exception:
                pass
            if achieved is None:
                basis = f"no measured bytes for this build ({pmc_state}): frac is null — run tools/gpu_profile.sh to refresh profiles/latest_pmc.json"
        out = {
            "metric": "env-steps/sec (decision events/sec), CIM global_trade.22p",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32+f64", "data": "synthetic",
            "repeats": R, "value_min": min(vals), "value_max": max(vals), "values": vals,
            "value_note": f"the timed window of exactly {args.steps} steps was run {R} times back to back; value / ms_per_step are the median window",
            "gpu_seconds_total": sum(v for v in gpu_busy_s.values() if v), "gpu_seconds": gpu_busy_s,
            "ranks": {"env_steps": [float(p[med, 1]) for p in per_rank], "seconds": [float(p[med, 0]) for p in per_rank],
                      "what": "each rank's env-steps and wall time of the median window: value = sum(env_steps) / max(seconds)"},
            "config": {"workload": f"CIM {args.topology}, {n} envs/GPU x {world} GPU, durations {sim_durations}, "
                                   f"{'random legal agent' if args.policy == 'random' else 'per-port dueling DQN (f32 MFMA, greedy) + CIMEnvSampler state shaping (mrx_cim_dqn_act)'} on device, ports + deciding-vessel snapshot slices {'off' if args.no_query else 'every step (' + args.obs + ')'}",
                       "envs_per_gpu": n, "groups_per_gpu": G, "step_mode": engines[0].step_mode, "specialized_kernels": bool(engines[0].specialized), "code_object_key": code_key, "code_object_sha16": getattr(engines[0], "code_object_sha16", None), "hip_graphs": bool(args.graphs), "agent": ("random legal agent answered inside the step kernel (mrx_cim_set_device_agent): one launch per batch step and group" if agent == "fused" else "mrx_cim_random_policy launch before every step"), "envs_per_launch": ng, "ring_slots": args.ring,
                       "parallelism": f"env-shard x{world} (no data-path collective); {G} independent groups per GPU on separate HIP streams",
                       "order_table": bool(engines[0].layout.order_table_on), "reset_ms_whole_batch": reset_ms, "host_enqueue_ms_per_step": t_issued / args.steps * 1e3,
                       "trajectory_gather_ms_32_steps": gather_ms, "trajectory_gather_bytes_per_rank": gather_bytes, "trajectory_gather_every": gather_every,
                       "untimed_preroll_steps": preroll_steps, "mean_tick_at_window_start": tick_mean0,
                       "mean_ticks_per_env_step": tbar, "envs_finished_in_window": n_done, "env_status_errors": status_bad},
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": None if achieved is None else achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": pmc_src, "basis": basis,
                         "bytes_per_env_step": None if traffic is None else traffic / ng,
                         "kernel_ms": step_kernel_ms, "launches_in_flight": in_flight, "env_steps_per_launch": ng, "launches_per_step": G,
                         "survey_formula": {"bytes_per_env_step": b_step, "bytes_per_launch": alg_per_launch, "GBps_if_the_engine_moved_them": alg_gbps,
                                            "measured_over_formula_bytes": None if traffic is None else traffic / alg_per_launch,
                                            "note": "SURVEY.md 8(d)'s fixed formula (reference dtypes, every snapshot copy counted).  The engine moves a fraction of these "
                                                    "bytes (aliased pre-decision snapshot, compact frame, HBM-direct fast path), so this is a statement about eliminated "
                                                    "copies, NOT a bandwidth figure, and it never feeds `frac`"}},
        }
        if ceiling is not None:
            # value / (env-steps/s the no-compute pattern reaches) — both whole-batch rates on one GPU
            out["roofline"]["pattern_ceiling"] = {"env_steps_per_s": ceiling["env_steps_per_s"], "GBps": ceiling["GBps"], "frac_of_peak": ceiling["GBps"] / HBM_PEAK_GBPS,
                                                  "value_over_ceiling": (value / world) / ceiling["env_steps_per_s"], "source": ceiling.get("source"),
                                                  "what": "tools/hbm_pattern_bench: the step kernel's loads / stores per env-step (same grid, LDS reservation, piece sizes, "
                                                          "sorted launch, 3 streams) with no simulation work"}
        if episode is not None:
            ep_s = float(ep_t.item())
            out["value_end_to_end"] = ep_steps_all / ep_s
            out["end_to_end"] = {"definition": "env-steps of one COMPLETE episode of every env / wall time from the first reset launch until the slowest env is done (reset included)",
                                 "env_steps": ep_steps_all, "seconds": ep_s, "reset_ms_synchronised": reset_ms, "reset": "per group on its own stream, overlapped with the other groups' first steps", "batch_steps": episode["batch_steps"],
                                 "durations": engines[0].durations}
        if sustained is not None:
            # (per rank: every rank runs the same window on its own GPU; the line carries rank 0's, scaled by the world size)
            out["value_sustained"] = sustained["env_steps"] * world / sustained["seconds"]
            out["sustained"] = dict(sustained, definition="env-steps of %d staggered groups running whole episodes back to back for %d batch steps each (resets included, "
                                    "a group is reset as soon as its polled done flag says so; envs that finish early idle until their group's slowest one is through) / wall time"
                                    % (G, sustained["batch_steps_per_group"]), over_steady_state=out["value_sustained"] / value)
            out["config"]["value_sustained"] = out["value_sustained"]
        if episode is not None:
            out["config"]["value_end_to_end"] = out["value_end_to_end"]
        if parity is not None:
            out["parity"] = parity
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args.topology, args.durations, args.cpu_seconds)
            # the reference's own Env.step / VectorEnv, timed live on this box's host cores from the shipped build (oracle/_ref)
            ref = cpu_baseline_reference(args.topology, args.durations, args.cpu_seconds)
            if ref is None or "error" in ref:
                # no built reference reachable: fall back to the committed record of the build container, and say so
                err = None if ref is None else ref.get("error")
                try:
                    with open(os.path.join(REPO, "profiles", "cpu_reference_baseline.json")) as fp:
                        ref = json.load(fp)
                    ref["measured"] = "NOT in this run: embedded record (profiles/cpu_reference_baseline.json)" + (f"; the live leg failed: {err}" if err else "")
                    if ref.get("topology") != args.topology or ref.get("durations") != args.durations:
                        ref = None
                except Exception:
                    ref = None
            if ref is not None:
                out["cpu_baseline_reference"] = ref
        if qnet is not None:
            # the policy's own roofline: exact-f32 MFMA (MI355X_MICROARCH.md: 157.3 TFLOP/s), algorithmic flops = 2 x sum(in x out)
            # of the example's real layer sizes x the deciding envs of one launch; mrx_cim_dqn_act = bin + forward kernels
            fl = 2.0 * sum(a * b for a, b in ((171, 256), (256, 128), (128, 64), (64, 32), (32, 128), (32, 128), (128, 21), (128, 1)))
            tf = fl * (resolved / max(args.steps * world * G, 1)) / (policy_ms * 1e-3) / 1e12
            out["roofline_policy"] = {"bound": "mfma", "kernel": "mrx_k_cim_dqn_mlp16 (+ mrx_k_cim_dqn_prep: binning + state rows)", "achieved": tf, "peak": 157.3,
                                      "unit": "TFLOP/s", "frac": tf / 157.3, "dtype": "f32 (v_mfma_f32_16x16x4_f32)", "kernel_ms": policy_ms,
                                      "flops_per_env": fl, "note": "launch latency bound: ~190 32-env tiles per launch on 256 CUs, sharing them with the other groups' step kernels"}
        return out
    return None


if __name__ == "__main__":
    main()
