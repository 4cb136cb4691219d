"""Phase timing of mrx_k_cim_dqn_forward (s_memtime deltas per workgroup).  Needs the profiling build:
   cd maro_amd/csrc && hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fPIC -shared -DMRX_DQN_PROFILE \
       -o libmaro_amd.so cim_engine.hip cb_engine.hip        (rebuild without the flag afterwards)"""
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
fused.act(actions, n_actions, q=q)
torch.cuda.synchronize()
t = q.view(-1)[: 16 * (n // 32 + 23)].view(-1, 16)
t = t[t[:, 0] > 0]
names = ["row lookups", "state gather"] + [f"layer {i}" for i in range(6)] + ["argmax+translate"]
print("workgroups", t.shape[0], "cycles (100 MHz s_memtime ticks x ?):")
for i, nm in enumerate(names):
    print(f"  {nm:18s} mean {float(t[:, i].mean()):9.0f}  max {float(t[:, i].max()):9.0f}")
