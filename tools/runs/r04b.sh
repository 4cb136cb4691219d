#!/bin/bash
# round 4, call B: the device-resident collection loop (tests + group counts + calls of 20 / 64 steps), citi_bike env groups on streams
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r04b; mkdir -p $O
timeout 600 python -m pytest tests/test_sampler.py tests/test_gpu_dqn.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -15 $O/pytest.log
C="--policy dqn --collect --ring 8 --envs 8192 --no-cpu --repeats 3"
for g in 1 2 3 4; do timeout 200 python bench.py $C --groups $g --parity-envs 0 --steps 64 --warmup 16 > $O/collect_g$g.json 2> $O/collect_g$g.err; echo "collect g$g rc $?"; done
for g in 2 3; do timeout 200 python bench.py $C --groups $g --parity-envs 6 --steps 20 --warmup 5 > $O/collect20_g$g.json 2> $O/collect20_g$g.err; echo "collect20 g$g rc $?"; done
MRX_SAMPLER_V2=0 timeout 200 python bench.py $C --groups 3 --parity-envs 0 --steps 64 --warmup 16 > $O/collect_old_g3.json 2> $O/collect_old_g3.err
B="--scenario citi_bike --no-cpu --bounded-budget 0 --repeats 3"
for g in 1 2 3 4; do timeout 200 python bench.py $B --cb-groups $g --steps 200 --warmup 50 > $O/cb_toy_g$g.json 2> $O/cb_toy_g$g.err; echo "cb toy g$g rc $?"; done
for g in 1 2 3; do timeout 300 python bench.py $B --cb-groups $g --topology city.800s --envs 4096 --durations 2880 --steps 900 --warmup 300 --step-budget 64 --parity-envs 0 > $O/cb_city_g$g.json 2> $O/cb_city_g$g.err; echo "cb city g$g rc $?"; done
for f in $O/collect*.json $O/cb_*.json; do python - "$f" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]/1e6,2), "M", round(d["value_min"]/1e6,1), round(d["value_max"]/1e6,1), "ms", round(d["ms_per_step"],4), "parity", (d.get("parity") or {}).get("ok"), d["config"].get("host_enqueue_ms_per_step"))
except Exception as e: print(sys.argv[1], "ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-600:])
P
done
