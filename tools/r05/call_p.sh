#!/bin/bash
O=gpurun_out/r05p; mkdir -p $O
export PYTHONPATH=$PWD
timeout 300 python bench.py --scenario citi_bike --no-cpu --steps 400 --warmup 100 --repeats 3 --parity-envs 8 > $O/cb_new.json 2> $O/cb_new.err
MRX_CB_LANES=16 timeout 300 python bench.py --scenario citi_bike --no-cpu --steps 400 --warmup 100 --repeats 3 --parity-envs 8 > $O/cb_new_lanes16.json 2> $O/cb_new_lanes16.err
timeout 300 python - > $O/lanes_api.log 2>&1 <<'P'
import numpy as np, torch, time
from maro_amd.citi_bike.engine import CitiBikeBatchEngine
n=4096
def run(eng, k=300):
    a = torch.zeros((n,1,3), dtype=torch.int32, device="cuda"); na = torch.zeros((n,), dtype=torch.int32, device="cuda")
    eng.step()
    for i in range(1, 50): eng.random_policy(i, a, na); eng.step(a, na)
    torch.cuda.synchronize(); t=time.time()
    for i in range(50, 50+k): eng.random_policy(i, a, na); d,m,dn = eng.step(a, na)
    torch.cuda.synchronize(); return n*k/(time.time()-t)/1e6, m.clone()
e1 = CitiBikeBatchEngine("toy.3s_4t", n, durations=44000, snapshot_resolution=10, max_snapshots=16, specialize=True, seeds=np.arange(n)+1)
r1, m1 = run(e1)
e2 = CitiBikeBatchEngine("toy.3s_4t", n, durations=44000, snapshot_resolution=10, max_snapshots=16, specialize=True, seeds=np.arange(n)+1)
e2.set_lanes_per_wave(16)
r2, m2 = run(e2)
e2.set_lanes_per_wave(0)
print("auto", round(r1,1), "M; lanes 16 (runtime-shift build)", round(r2,1), "M; metrics equal", bool((m1==m2).all()), "code objects", e1.code_object_key, e2.code_object_key)
P
cat $O/lanes_api.log | grep -v amdgpu
timeout 900 python -m pytest tests/test_gpu_citi_bike.py tests/test_gpu_citi_bike_api.py tests/test_gpu_specialized.py -m gpu -x -q > $O/pytest_cb.log 2>&1; echo "pytest rc $?" >> $O/pytest_cb.log
python - <<P
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d["value"]/1e6,2), "M", round(d["ms_per_step"]*1e3,1), "us", "parity", (d.get("parity") or {}).get("ok"), d["config"].get("specialized_kernels"))
    except Exception as e: print(f, "FAILED", e, open(f.replace('.json','.err')).read()[-300:])
P
tail -3 $O/pytest_cb.log
