#!/bin/bash
# round 4, call H: XCD-aware tile order of the DQN forward kernel, A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r04h; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_dqn.py tests/test_sampler.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
C="--policy dqn --collect --ring 8 --envs 8192 --no-cpu --repeats 3 --parity-envs 0 --steps 64 --warmup 16 --groups 2"
for x in 0 1; do for t in 16 32; do MRX_DQN_XCD=$x MRX_DQN_TILE=$t timeout 200 python bench.py $C > $O/collect_x${x}_t$t.json 2> $O/collect_x${x}_t$t.err; echo "collect x$x t$t rc $?"; done; done
for x in 0 1; do MRX_DQN_XCD=$x timeout 200 python bench.py $C --groups 1 > $O/collect_x${x}_g1.json 2> $O/collect_x${x}_g1.err; done
for x in 0 1; do MRX_DQN_XCD=$x timeout 200 python bench.py --policy dqn --ring 8 --envs 16384 --no-cpu --parity-envs 0 --secondary 0 --steps 200 --repeats 3 > $O/dqn_x$x.json 2> $O/dqn_x$x.err; done
for f in $O/collect_*.json $O/dqn_*.json; do python - "$f" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]/1e6,2), "M", round(d["value_min"]/1e6,1), round(d["value_max"]/1e6,1), "ms", round(d["ms_per_step"],4), "act_ms", d.get("roofline_policy",{}).get("kernel_ms"))
except Exception as e: print(sys.argv[1], "ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-600:])
P
done
