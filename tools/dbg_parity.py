import sys; sys.path.insert(0,'.')
import numpy as np, torch
from maro_amd.cim.engine import CimBatchEngine
from oracle.cim_oracle import CimOracle
topo = "global_trade.22p_l0.8"
n = 5462
seeds = torch.arange(n, dtype=torch.int64) + 1
exp = {}
def oracle_first(seed):
    if seed not in exp:
        o = CimOracle(topo, durations=1120, max_snapshots=4); o.set_seed(seed); o.reset(keep_seed=True)
        m, d, dn = o.step(None); exp[seed] = (m.copy(), d.copy())
    return exp[seed]
def check(eng, tag):
    dec, met, done = (x.cpu().numpy() for x in eng.step())
    torch.cuda.synchronize()
    bad = []
    for e in range(0, n, 7):
        m, d = oracle_first(e + 1)
        if not (np.array_equal(m, met[e]) and np.array_equal(d, dec[e])): bad.append(e)
    print(tag, "mismatching envs:", len(bad), bad[:20])
for spec, mode in ((True, 0), (False, 0), (True, 1)):
    eng = CimBatchEngine(topo, n, durations=1120, max_snapshots=4, max_actions=1, seeds=seeds, specialize=spec, step_mode=mode)
    eng.set_observation(["empty", "full", "on_shipper", "on_consignee", "booking", "shortage", "fulfillment"], ["empty", "full", "remaining_space"])
    check(eng, f"spec={spec} mode={mode} fresh")
    a = torch.zeros((n,1,4), dtype=torch.int32, device="cuda"); na = torch.zeros(n, dtype=torch.int32, device="cuda")
    for i in range(1, 300):
        eng.random_policy(i, a, na); eng.step(a, na)
    eng.reset(seeds)
    check(eng, f"spec={spec} mode={mode} after 300 steps + reset")
    st = torch.cuda.Stream(); eng.use_stream(st)
    for i in range(1, 300):
        eng.random_policy(i, a, na); eng.step(a, na)
    eng.reset(seeds)
    check(eng, f"spec={spec} mode={mode} bound stream, after 300 steps + reset")
    del eng
