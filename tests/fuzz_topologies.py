"""Randomised differential test (tooling + a small pytest slice): random CIM topologies within the engine's limits,
device source on the CPU wave emulator vs the C oracle, random actions.  `python tests/fuzz_topologies.py N [seed0]`
runs N cases; tests/test_emu_fuzz.py runs a fixed handful in the CPU suite."""
import copy
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))


def random_conf(rng):
    P = int(rng.randint(2, 9))
    names = [f"p{i}" for i in range(P)]
    R = int(rng.randint(1, 4))
    routes = {}
    for r in range(R):
        L = int(rng.randint(2, min(P, 6) + 1))
        pts = list(rng.choice(P, L, replace=False))
        if rng.rand() < 0.35 and L >= 3:      # visit a port twice
            pts.insert(int(rng.randint(2, L + 1)), pts[0])
            if pts[-1] == pts[0]:
                pts.append(int([p for p in range(P) if p != pts[0]][0]))
        routes[f"r{r}"] = [{"port_name": names[p], "distance_to_next_port": int(rng.randint(4, 40))} for p in pts]
    on_route = sorted({pt["port_name"] for pts in routes.values() for pt in pts})
    ports = {}
    prop = rng.dirichlet(np.ones(P))
    prop = np.floor(prop * 1000) / 1000
    prop[0] += round(1 - prop.sum(), 3)
    src = rng.dirichlet(np.ones(P))
    for i, n in enumerate(names):
        others = [m for m in names if m != n]
        k = int(rng.randint(0, min(len(others), 4) + 1))
        tg = list(rng.choice(others, k, replace=False)) if k else []
        tp = rng.dirichlet(np.ones(k)) if k else []
        noise_s = float(rng.choice([0, 0.02, 0.1])) if k else 0.0
        od = {"source": {"proportion": float(round(src[i], 3)) if k else 0.0, "noise": noise_s}}
        if k:
            od["targets"] = {t: {"proportion": float(round(tp[j], 3)), "noise": float(rng.choice([0, 0.05, 0.2]))} for j, t in enumerate(tg)}
        ports[n] = {"capacity": int(rng.randint(2000, 20000)),
                    "empty_return": {"buffer_ticks": int(rng.randint(0, 4)), "noise": int(rng.randint(0, 3))},
                    "full_return": {"buffer_ticks": int(rng.randint(0, 4)), "noise": int(rng.randint(0, 3))},
                    "initial_container_proportion": float(prop[i]), "order_distribution": od}
    V = int(rng.randint(1, 9))
    vol = int(rng.choice([1, 1, 2]))
    vessels = {}
    for v in range(V):
        r = f"r{int(rng.randint(0, R))}"
        p0 = routes[r][int(rng.randint(0, len(routes[r])))]["port_name"]
        vessels[f"v{v}"] = {"capacity": int(rng.randint(50 * vol, 2000)), "parking": {"duration": int(rng.randint(1, 4)), "noise": int(rng.randint(0, 2))},
                            "sailing": {"speed": int(rng.randint(6, 14)), "noise": int(rng.randint(0, 3))},
                            "route": {"route_name": r, "initial_port_name": p0}, "empty": int(rng.randint(0, 20))}
    total = int(rng.choice([2000, 10000, 50000]))
    nodes = sorted({0, 19} | set(int(x) for x in rng.randint(1, 19, 3)))
    return {"seed": int(rng.randint(0, 5000)), "load_cost_factor": 0.05, "dsch_cost_factor": 0.05,
            "container_usage_proportion": {"period": 20, "sample_nodes": [[x, float(round(rng.uniform(0.0, 0.08), 3))] for x in nodes],
                                           "sample_noise": float(rng.choice([0, 0.004, 0.02]))},
            "container_volumes": [vol], "order_generate_mode": str(rng.choice(["fixed", "fixed", "unfixed"])),
            "total_containers": total, "stop_number": [int(rng.randint(1, 5)), int(rng.randint(1, 5))],
            "ports": ports, "routes": routes, "vessels": vessels}


def run_case(case_seed, durations=70, backend=None):
    from tests.emu.emu import EmuBackend
    from tests.test_emu_synthetic import run_pair
    rng = np.random.RandomState(case_seed)
    conf = random_conf(rng)
    try:
        res, seed = int(rng.choice([1, 1, 2, 3, 5])), int(rng.randint(0, 10**6))
        start = int(rng.choice([0, 0, 1, 2, 3, 13]))   # start_tick > 0: departures before it never ran ("zombie" vessels); small ones keep some alive
        ring = None if rng.rand() < 0.6 else int(rng.randint(2, 9))   # a small snapshot ring: eviction order
        return run_pair(copy.deepcopy(conf), durations=durations, resolution=res, ring=ring, seed=seed, min_steps=0, backend=backend or EmuBackend,
                        start_tick=start, segments=2 if case_seed % 3 == 0 else 1)
    except Exception:
        import json
        print("FAILING CASE seed", case_seed, json.dumps(conf)[:2000])
        raise


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    ok = 0
    for s in range(s0, s0 + n):
        run_case(s)
        ok += 1
        if ok % 10 == 0:
            print(ok, "cases ok", flush=True)
    print("all", ok, "cases ok")
