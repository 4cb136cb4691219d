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

# reference-dtype frame bytes per env and targets per topology family
FRAME_BYTES = {"global_trade.22p": 15412, "toy.4p_ssdd": 886}
QUERY_ATTRS = ["empty", "full", "on_shipper", "on_consignee", "booking", "shortage", "fulfillment"]
VESSEL_QUERY_ATTRS = ["empty", "full", "remaining_space"]
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def init_dist(world, local_rank):
    """One process per GPU over RCCL (backend "nccl").  MRX_BENCH_BACKEND=gloo + MRX_BENCH_DEVICE=<i> are test hooks only: they
    let a 1-GPU box run the N>1 code path with every rank on the same device (RCCL refuses two ranks per GPU)."""
    import torch
    dev = torch.device(f"cuda:{os.environ.get('MRX_BENCH_DEVICE', local_rank)}")
    if world <= 1:
        return None, dev
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(dev)
    backend = os.environ.get("MRX_BENCH_BACKEND", "nccl")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    dist.barrier()
    return dist, dev


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
            "note": "reference Python Env.step measured in the build container: ~311 env-steps/s (BASELINE.md §2)"}


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


def bench_citi_bike(args):
    """BASELINE.json configs[3]: citi_bike toy.3s_4t, 4096 envs per GPU, shared trip table, per-env action seeds.
    One step = device policy -> mrx_cb_step (action + ticks until the next decision) -> stations snapshot slice."""
    import torch

    import __graft_entry__ as ge
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if rank == 0:
        ge.build()
    dist, dev = init_dist(world, local_rank)
    torch.cuda.set_device(dev)
    import numpy as np

    from maro_amd.citi_bike.engine import CitiBikeBatchEngine

    n, res = args.envs, 10
    topology = args.topology if args.topology != "global_trade.22p_l0.8" else "toy.3s_4t"
    durations = args.durations if args.durations != 1120 else 44000
    kw = dict(durations=durations, snapshot_resolution=res, max_snapshots=16, max_actions=1, device=dev, seeds=np.arange(n) + rank * n + 1)
    try:
        eng = CitiBikeBatchEngine(topology, n, specialize=bool(args.specialize), **kw)   # kernels compiled for this plan (cached in-tree)
    except (RuntimeError, OSError, subprocess.CalledProcessError) as e:
        if not args.specialize:
            raise
        print(f"bench: specialised kernels unavailable ({e}); using the generic ones", file=sys.stderr)
        eng = CitiBikeBatchEngine(topology, n, specialize=False, **kw)
    S = eng.data.n_stations
    actions = torch.zeros((n, 1, 3), dtype=torch.int32, device=dev)
    n_actions = torch.zeros((n,), dtype=torch.int32, device=dev)
    counter = torch.zeros((1,), dtype=torch.int64, device=dev)
    stations = torch.arange(S, dtype=torch.int32, device=dev)
    q_attrs = ["bikes", "shortage", "trip_requirement", "fulfillment", "capacity", "extra_cost", "min_bikes"]
    q_out = None if args.no_query else torch.empty((n, 1, S, len(q_attrs)), dtype=torch.float64, device=dev)

    def one_step(i):
        if i == 0:
            eng.step()
            return
        eng.random_policy(i, actions, n_actions, counter)
        eng.step(actions, n_actions)
        if q_out is not None:
            eng.query("stations", eng.decisions[:, 3:4], stations, q_attrs, out=q_out)

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    step_i = 0
    for _ in range(args.warmup):
        one_step(step_i)
        step_i += 1
    sync_all()
    counter.zero_()
    tick0 = eng.ticks.to(torch.int64).sum().item()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step(step_i)
        step_i += 1
    sync_all()
    dt = time.perf_counter() - t0
    resolved = int(counter.item())
    ticks_adv = eng.ticks.to(torch.int64).sum().item() - tick0
    n_done = int(eng.done.sum().item())
    status_bad = int((eng.status != 0).sum().item())
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(min(args.steps, 100))]
    for a, b in ev:
        eng.random_policy(step_i, actions, n_actions, None)
        a.record()
        eng.step(actions, n_actions)
        b.record()
        step_i += 1
    torch.cuda.synchronize(dev)
    step_kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
    # the one exchange step of a sharded rollout: 32 steps of (decision, action, metrics, done) per env gathered to the
    # learner rank over RCCL (maro_amd/cim/rollout.py::gather_to_learner); outside the timed env-step window
    gather_ms = None
    if dist is not None:
        from maro_amd.cim.rollout import gather_to_learner
        T = 32
        traj = {"decisions": torch.zeros((T, n, 8), dtype=torch.int32, device=dev), "actions": torch.zeros((T, n, 1, 3), dtype=torch.int32, device=dev),
                "metrics": torch.zeros((T, n, 3), dtype=torch.int64, device=dev), "done": torch.zeros((T, n), dtype=torch.uint8, device=dev)}
        for k in range(T):
            eng.random_policy(step_i, actions, n_actions, None)
            traj["decisions"][k], traj["actions"][k] = eng.decisions, actions
            eng.step(actions, n_actions)
            traj["metrics"][k], traj["done"][k] = eng.metrics, eng.done
            step_i += 1
        sync_all()
        tg = time.perf_counter()
        out = gather_to_learner(traj, dst=0)
        sync_all()
        gather_ms = (time.perf_counter() - tg) * 1e3
        if rank == 0:
            assert out["decisions"].shape[1] == n * world
    t_max = torch.tensor([dt], dtype=torch.float64, device=dev)
    tot = torch.tensor([float(resolved), float(ticks_adv), float(n_done), float(status_bad)], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    dt = float(t_max.item())
    resolved, ticks_adv, n_done, status_bad = (float(x) for x in tot.tolist())
    if rank == 0:
        tbar = ticks_adv / max(resolved, 1.0)
        F = S * 48 + 4 * S * S                       # reference-dtype frame bytes (180 B for 3 stations)
        n_trips = float((eng.data.trip_tick < durations).sum())
        b_step = (3.0 + tbar / res) * F + 20.0 * (n_trips / durations) * tbar + 40.0   # SURVEY.md §8(d) general form
        achieved = b_step * n / (step_kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "env-steps/sec (decision events/sec), citi_bike toy.3s_4t",
            "value": resolved / dt, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32+f64", "data": "synthetic",
            "config": {"workload": f"citi_bike {topology}, {n} envs/GPU x {world} GPU, durations {durations}, resolution {res}, "
                                   f"device policy, stations snapshot slice {'off' if args.no_query else 'every step'}",
                       "envs_per_gpu": n, "specialized_kernels": bool(eng.specialized), "ring_slots": 16, "parallelism": f"env-shard x{world} (no data-path collective)",
                       "trajectory_gather_ms_32_steps": gather_ms, "mean_ticks_per_env_step": tbar, "envs_finished_in_window": n_done, "env_status_errors": status_bad},
            "roofline": {"bound": "hbm", "kernel": "mrx_k_cb_step", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "kernel_ms": step_kernel_ms,
                         "algorithmic_bytes_per_env_step": b_step, "env_steps_per_launch": n,
                         "note": "latency-bound by construction (SURVEY.md §8d: ~1 KB per env-step); the roofline fraction is judged on the CIM 22p workload"},
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline_citi_bike(topology, min(durations, 1440), res, args.cpu_seconds)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


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
    ap.add_argument("--obs", default="fused", choices=["fused", "query"],
                    help="how the per-step ports / deciding-vessel snapshot slices are produced: fused into the step kernel, or by mrx_cim_query")
    ap.add_argument("--graphs", type=int, default=0, help="1: capture one step per group in a hipGraph and replay it (cim)")
    ap.add_argument("--groups", type=int, default=3, help="independent env groups per GPU, each on its own HIP stream (cim)")
    ap.add_argument("--step-mode", type=int, default=0, help="launch form of mrx_cim_step (mrx_cim_set_step_mode): 0 best available, "
                    "1 unsorted, 2 sorted, 3 persistent pipelined")
    ap.add_argument("--topology", default="global_trade.22p_l0.8")
    ap.add_argument("--durations", type=int, default=1120)
    ap.add_argument("--specialize", type=int, default=1, help="1: step with kernels compiled for this exact plan (maro_amd/cim/specialize.py; "
                    "built by __graft_entry__.build() for the default workload, else ~3 s of hipcc at engine creation); 0: generic kernels")
    ap.add_argument("--ring", type=int, default=4, help="snapshot ring slots per env (max_snapshots)")
    ap.add_argument("--no-query", action="store_true", help="skip the per-step snapshot slice")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    if args.envs is None:
        args.envs = 16384 if args.scenario == "cim" else 4096
    if args.scenario == "citi_bike":
        return bench_citi_bike(args)

    import torch

    import __graft_entry__ as ge
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if rank == 0:
        ge.build()
    dist, dev = init_dist(world, local_rank)
    torch.cuda.set_device(dev)

    from maro_amd.cim.engine import CimBatchEngine

    # The per-GPU batch is split into G independent groups, each with its own engine and HIP stream: a step kernel
    # ends with a tail of long (ticking) waves while most of the chip is already idle, and the next group's kernel
    # fills exactly that tail.  Envs never interact, so this is pure scheduling.
    # the episode must outlast the run (finished envs would idle): ~0.45 ticks per env-step on 22p, so only very long runs
    # (more than ~2300 steps in total) stretch the nominal 1120 ticks
    sim_durations = max(args.durations, int(0.55 * (args.warmup + args.steps + min(args.steps, 100) + 10)) + 64)
    n, G = args.envs, max(1, args.groups)
    sizes = [n // G + (1 if g < n % G else 0) for g in range(G)]   # group sizes differ by at most one env
    offs = [sum(sizes[:g]) for g in range(G)]
    engines, streams, bufs = [], [], []
    for g in range(G):
        ng = sizes[g]
        seeds = torch.arange(ng, dtype=torch.int64) + rank * n + offs[g] + 1
        kw = dict(durations=sim_durations, max_snapshots=max(args.ring, 8) if args.policy == "dqn" else args.ring, max_actions=1,
                  device=dev, seeds=seeds)  # dqn: the look-back window must fit the ring
        try:
            eng = CimBatchEngine(args.topology, ng, specialize=bool(args.specialize), step_mode=args.step_mode, **kw)
        except (RuntimeError, OSError, subprocess.CalledProcessError) as e:   # no hipcc and not in the cache: generic kernels
            if not args.specialize:
                raise
            print(f"bench: specialised kernels unavailable ({e}); using the generic ones", file=sys.stderr)
            eng = CimBatchEngine(args.topology, ng, specialize=False, step_mode=args.step_mode, **kw)
        engines.append(eng)
        streams.append(torch.cuda.Stream(device=dev) if G > 1 else torch.cuda.current_stream(dev))
        bufs.append(dict(actions=torch.zeros((ng, 1, 4), dtype=torch.int32, device=dev),
                         n_actions=torch.zeros((ng,), dtype=torch.int32, device=dev),
                         counter=torch.zeros((1,), dtype=torch.int64, device=dev),
                         q_ports=None if (args.no_query or args.obs == "fused") else torch.empty((ng, 1, engines[0].topo.n_ports, len(QUERY_ATTRS)), dtype=torch.float64, device=dev),
                         q_vessel=None if (args.no_query or args.obs == "fused") else torch.empty((ng, 1, 1, len(VESSEL_QUERY_ATTRS)), dtype=torch.float64, device=dev)))
        if args.obs == "fused" and not args.no_query and args.policy != "dqn":
            # the same two slices, written by the step kernel itself (mrx_cim_set_observation) instead of two more launches
            bufs[-1]["obs"] = eng.set_observation(QUERY_ATTRS, VESSEL_QUERY_ATTRS)
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
    # Env.reset for the whole batch (route unrolling, order proportion and — with the order table — every order of the
    # episode are generated on the device here, outside the timed step loop): reported next to the step rate
    t_r = time.perf_counter()
    for g, eng in enumerate(engines):
        with torch.cuda.stream(streams[g]):
            eng.reset(torch.arange(sizes[g], dtype=torch.int64) + rank * n + offs[g] + 1)
    torch.cuda.synchronize(dev)
    reset_ms = (time.perf_counter() - t_r) * 1e3

    graphs = [None] * G

    for eng, st in zip(engines, streams):
        eng.use_stream(st)   # every engine call goes to its group's stream without a per-call stream switch

    def one_step(i, g, timing=None):
        eng, b, st = engines[g], bufs[g], streams[g]
        if i == 0:
            eng.step()  # first step of the episode: action=None
            return
        if graphs[g] is not None and timing is None:
            with torch.cuda.stream(st):
                graphs[g].replay()  # policy -> step -> snapshot slices, captured once (hipGraph)
            return
        if qnet is not None:
            # mrx_cim_dqn_act: state gather + MFMA MLP + argmax + translation
            if timing is not None:
                timing[2].record(st)
            qnet[g].act(b["actions"], b["n_actions"], counter=b["counter"] if timing is None else None)
        else:
            eng.random_policy(-1 if args.graphs else i, b["actions"], b["n_actions"], b["counter"] if timing is None else None)
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

    step_i = 0
    for _ in range(args.warmup):
        for g in range(G):
            one_step(step_i, g)
        step_i += 1
    sync_all()
    if args.graphs:
        # the launch-bound inner loop (4 small launches per group and step) is captured once per group and replayed
        for g in range(G):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=streams[g]):
                eng, b = engines[g], bufs[g]
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
    for b in bufs:
        b["counter"].zero_()
    tick0 = total_ticks()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for g in range(G):
            one_step(step_i, g)
        step_i += 1
    t_issued = time.perf_counter() - t0   # host time to enqueue every launch of the window (the loop never waits for the GPU)
    sync_all()
    dt = time.perf_counter() - t0
    # decisions answered inside the timed window (K policy calls per group in the window)
    resolved = sum(int(b["counter"].item()) for b in bufs)
    ticks_adv = total_ticks() - tick0
    n_done = sum(int(e.done.sum().item()) for e in engines)
    status_bad = sum(int((e.status != 0).sum().item()) for e in engines)

    # ---- dominant kernel (mrx_k_cim_step) timed live with HIP events, each pair on the stream of its launch, in
    # the same interleaved schedule as the timed loop
    reps = min(args.steps, 100)
    ev = [[tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(G)] for _ in range(reps)]
    sync_all()
    for r in range(reps):
        for g in range(G):
            one_step(step_i, g, timing=ev[r][g])
        step_i += 1
    torch.cuda.synchronize(dev)
    durs = [a.elapsed_time(b) for row in ev for a, b, _ in row]
    policy_ms = sum(c.elapsed_time(a) for row in ev for a, _, c in row) / len(durs) if qnet is not None else None
    step_kernel_ms = sum(durs) / len(durs)                      # mean duration of one launch (ng envs)
    span_ms = max([ev[0][g][0].elapsed_time(ev[-1][g2][1]) for g in range(G) for g2 in range(G)]) if G > 1 else sum(durs)
    in_flight = max(1.0, sum(durs) / span_ms) if G > 1 else 1.0    # mean number of step kernels running concurrently

    # the one exchange step of a sharded rollout (north_star: RCCL "only to gather trajectories to the learner"): 32 steps of
    # (decision, action, metrics, done) per env from every rank to rank 0 (maro_amd/cim/rollout.py::gather_to_learner);
    # outside the timed env-step window
    gather_ms = None
    if dist is not None:
        from maro_amd.cim.rollout import gather_to_learner
        T = 32
        traj = {"decisions": torch.zeros((T, n, 8), dtype=torch.int32, device=dev), "actions": torch.zeros((T, n, 1, 4), dtype=torch.int32, device=dev),
                "metrics": torch.zeros((T, n, 3), dtype=torch.int64, device=dev), "done": torch.zeros((T, n), dtype=torch.uint8, device=dev)}
        for k in range(T):
            for g in range(G):
                one_step(step_i, g)
                with torch.cuda.stream(streams[g]):
                    sl = slice(offs[g], offs[g] + sizes[g])
                    traj["decisions"][k, sl], traj["actions"][k, sl] = engines[g].decisions, bufs[g]["actions"]
                    traj["metrics"][k, sl], traj["done"][k, sl] = engines[g].metrics, engines[g].done
            step_i += 1
        sync_all()
        tg = time.perf_counter()
        gathered = gather_to_learner(traj, dst=0)
        torch.cuda.synchronize(dev)
        gather_ms = (time.perf_counter() - tg) * 1e3
        if rank == 0:
            assert gathered["decisions"].shape[1] == n * world

    t_max = torch.tensor([dt], dtype=torch.float64, device=dev)
    tot = torch.tensor([float(resolved), float(ticks_adv), float(n_done), float(status_bad)], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    dt = float(t_max.item())
    resolved, ticks_adv, n_done, status_bad = (float(x) for x in tot.tolist())

    if rank == 0:
        value = resolved / dt
        tbar = ticks_adv / max(resolved, 1.0)  # mean ticks advanced per env-step
        F = frame_bytes(topo)
        b_step = (3.0 + tbar) * F + 4.0 * topo.n_targets * tbar + 40.0
        ng = n / G                                  # mean env-steps per launch
        bytes_per_launch = b_step * ng
        # one launch moves bytes_per_launch in step_kernel_ms, and `in_flight` launches (one per group/stream) overlap
        achieved = bytes_per_launch / (step_kernel_ms * 1e-3) / 1e9 * in_flight
        traffic = None  # HBM bytes per launch from the committed PMC passes of this same workload (profiles/)
        try:
            with open(os.path.join(REPO, "profiles", "latest_pmc.json")) as fp:
                pmc = json.load(fp)
            if pmc["topology"] == args.topology and abs(pmc["envs_per_launch"] - ng) <= 1:
                traffic = (2.0 * pmc["fetch_size_kib"] + pmc["write_size_kib"]) * 1024.0
        except Exception:
            pass
        out = {
            "metric": "env-steps/sec (decision events/sec), CIM global_trade.22p",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32+f64", "data": "synthetic",
            "config": {"workload": f"CIM {args.topology}, {n} envs/GPU x {world} GPU, durations {sim_durations}, "
                                   f"{'random legal agent' if args.policy == 'random' else 'per-port dueling DQN (f32 MFMA, greedy) + CIMEnvSampler state shaping (mrx_cim_dqn_act)'} on device, ports + deciding-vessel snapshot slices {'off' if args.no_query else 'every step (' + args.obs + ')'}",
                       "envs_per_gpu": n, "groups_per_gpu": G, "step_mode": engines[0].step_mode, "specialized_kernels": bool(engines[0].specialized), "hip_graphs": bool(args.graphs), "envs_per_launch": ng, "ring_slots": args.ring,
                       "parallelism": f"env-shard x{world} (no data-path collective); {G} independent groups per GPU on separate HIP streams",
                       "order_table": bool(engines[0].layout.order_table_on), "reset_ms_whole_batch": reset_ms, "host_enqueue_ms_per_step": t_issued / args.steps * 1e3, "trajectory_gather_ms_32_steps": gather_ms,
                       "mean_ticks_per_env_step": tbar, "envs_finished_in_window": n_done, "env_status_errors": status_bad},
            "roofline": {"bound": "hbm", "kernel": ("mrx_k_cim_step_tab" if engines[0].layout.order_table_on else "mrx_k_cim_step") + ("_obs" if bufs[0].get("obs") is not None else ""), "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "traffic_GBps": None if traffic is None else traffic / (step_kernel_ms * 1e-3) / 1e9 * in_flight,
                         "traffic_source": "profiles/latest_pmc.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes per launch)",
                         "algorithmic_bytes_per_launch": bytes_per_launch, "kernel_ms": step_kernel_ms, "launches_in_flight": in_flight,
                         "definition": "achieved = algorithmic_bytes_per_launch / kernel_ms x launches_in_flight (mean number of overlapping step kernels, one per group stream); algorithmic bytes are SURVEY.md 8(d)'s fixed per-env-step formula (reference dtypes, every snapshot copy counted), so frac can exceed 1: the engine moves fewer real bytes (traffic, traffic_GBps)",
                         "algorithmic_bytes_per_env_step": b_step, "env_steps_per_launch": ng},
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args.topology, args.durations, args.cpu_seconds)
        if qnet is not None:
            # the policy's own roofline: exact-f32 MFMA (MI355X_MICROARCH.md: 157.3 TFLOP/s), algorithmic flops = 2 x sum(in x out)
            # of the example's real layer sizes x the deciding envs of one launch; mrx_cim_dqn_act = bin + forward kernels
            fl = 2.0 * sum(a * b for a, b in ((171, 256), (256, 128), (128, 64), (64, 32), (32, 128), (32, 128), (128, 21), (128, 1)))
            tf = fl * (resolved / max(args.steps * world * G, 1)) / (policy_ms * 1e-3) / 1e12
            out["roofline_policy"] = {"bound": "mfma", "kernel": "mrx_k_cim_dqn_forward (+ mrx_k_cim_dqn_bin)", "achieved": tf, "peak": 157.3,
                                      "unit": "TFLOP/s", "frac": tf / 157.3, "dtype": "f32 (v_mfma_f32_16x16x4_f32)", "kernel_ms": policy_ms,
                                      "flops_per_env": fl, "note": "launch latency bound: ~190 32-env tiles per launch on 256 CUs, sharing them with the other groups' step kernels"}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
