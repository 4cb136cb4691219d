"""Phase timing of mrx_k_cim_dqn_mlp16 (s_memtime deltas per workgroup).  Needs the profiling build, kept apart from the product:
   (cd maro_amd/csrc && hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fPIC -shared -DMRX_DQN_PROFILE \
       -o /tmp/libmaro_amd_prof.so cim_engine.hip cb_engine.hip) && MARO_AMD_LIB=/tmp/libmaro_amd_prof.so python tools/dqn_phase_profile.py [n_envs]
   MRX_DQN_TILE=16|32 selects the tile rows.  Prints the phases in s_memtime ticks and, from the event-timed duration of the same
   launch, what a tick is worth."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maro_amd.cim.engine import CimBatchEngine
from maro_amd.cim.policy import FusedPerPortDQN, random_chains
from maro_amd.cim.sampler import CimBatchSampler

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5461
eng = CimBatchEngine("global_trade.22p_l0.8", n, durations=1120, max_snapshots=8, order_table=1, seeds=torch.arange(n, dtype=torch.int64) + 1)
fused = FusedPerPortDQN(eng, random_chains(22, CimBatchSampler(eng).state_dim, 21, seed=0))
actions = torch.zeros((n, 1, 4), dtype=torch.int32, device="cuda"); n_actions = torch.zeros((n,), dtype=torch.int32, device="cuda")
q = torch.zeros((n + 4096 + 8 * (n // 16 + 64), 21), dtype=torch.float32, device="cuda")   # profile rows (16 floats per workgroup) behind the n q rows
eng.step()
for i in range(60):
    fused.act(actions, n_actions, q=q)
    eng.step(actions, n_actions)
q.zero_()
if os.environ.get("MRX_DQN_PROFILE_WARM"):   # the measured call right after an identical one: the weights are L2-warm, nothing ran in between
    fused.act(actions, n_actions, q=q)
    q.zero_()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
torch.cuda.synchronize()
ev[0].record()
fused.act(actions, n_actions, q=q)
ev[1].record()
torch.cuda.synchronize()
print(f"n_envs {n}, tile rows {os.environ.get('MRX_DQN_TILE', 'default')}: bin + forward launch = {ev[0].elapsed_time(ev[1]) * 1e3:.1f} us (events)")
grid = int(os.environ.get("MRX_DQN_GRID", 0)) or ((n + 15) // 16 + 22 + 7) // 8 * 8 + 8
t = q.view(-1)[n * 21: n * 21 + 16 * grid].view(-1, 16)
t = t[t[:, 0] > 0]
names = ["row loads"] + [f"layer {i}" for i in range(6)] + ["argmax+translate"]
print("workgroups", t.shape[0], "s_memtime ticks; sum of the phase means =", float(t[:, :8].mean(dim=0).sum()))
for i, nm in enumerate(names):
    print(f"  {nm:18s} mean {float(t[:, i].mean()):9.0f}  max {float(t[:, i].max()):9.0f}")
print(f"  marked span: {float(t[:, 14].mean()):.0f} s_memtime ticks in {float(t[:, 15].mean()) * 10:.0f} ns (s_memrealtime) -> {float(t[:, 14].sum() / (t[:, 15].sum() * 10)):.3f} ticks per ns = shader GHz under this launch")
tl = q.view(-1)[n * 21 + grid * 16: n * 21 + grid * 16 + grid * 256].view(grid, 4, 64)
tl = tl[(q.view(-1)[n * 21: n * 21 + grid * 16].view(grid, 16)[:, 0] > 0)]
full = tl[tl[:, 0, 8] > 0] if tl.shape[0] else tl
print("per-wave timelines (mean over workgroups, ticks since the first mark; -1: no stamp):")
for wv in range(4):
    m = tl[:, wv, :].mean(dim=0)
    print(f"  wave {wv}: " + " ".join(f"{float(v):.0f}" for v in m if float(v) >= 0))
if os.environ.get("MRX_DQN_PROFILE_DUMP"):
    import numpy as np
    a = t.cpu().numpy()
    t0 = a[:, 12].min()
    print("per workgroup (sorted by start): start_us dur_us port rows | phases")
    for i in np.argsort(a[:, 12]):
        print(f"  {(a[i, 12] - t0) / 100:7.2f} {a[i, 15] / 100:7.2f}  p{int(a[i, 10]) // 64:2d} r{int(a[i, 10]) % 64:2d} | " + " ".join(f"{int(v):6d}" for v in a[i, :8]))
