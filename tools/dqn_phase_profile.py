"""Phase timing of mrx_k_cim_dqn_forward (s_memtime deltas per workgroup).  Needs the profiling build, kept apart from the product:
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
q = torch.zeros((max(n, 4096), 21), dtype=torch.float32, device="cuda")
eng.step()
for i in range(60):
    fused.act(actions, n_actions, q=q)
    eng.step(actions, n_actions)
q.zero_()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
torch.cuda.synchronize()
ev[0].record()
fused.act(actions, n_actions, q=q)
ev[1].record()
torch.cuda.synchronize()
print(f"n_envs {n}, tile rows {os.environ.get('MRX_DQN_TILE', 'default')}: bin + forward launch = {ev[0].elapsed_time(ev[1]) * 1e3:.1f} us (events)")
t = q.view(-1)[: 16 * (n // 16 + 23)].view(-1, 16)
t = t[t[:, 0] > 0]
names = ["row lookups", "state gather"] + [f"layer {i}" for i in range(6)] + ["argmax+translate"]
print("workgroups", t.shape[0], "s_memtime ticks; sum of the phase means =", float(t[:, :9].mean(dim=0).sum()))
for i, nm in enumerate(names):
    print(f"  {nm:18s} mean {float(t[:, i].mean()):9.0f}  max {float(t[:, i].max()):9.0f}")
