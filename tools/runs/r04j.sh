#!/bin/bash
# round 4, call J: scope-rows observation in the wave kernels (city.800s), citi_bike GPU tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r04j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_citi_bike.py tests/test_gpu_citi_bike_api.py tests/test_gpu_specialized.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -6 $O/pytest.log
B="--scenario citi_bike --no-cpu --bounded-budget 0 --repeats 3"
C="--topology city.800s --envs 4096 --durations 2880 --steps 900 --warmup 300 --step-budget 64"
timeout 300 python bench.py $B $C > $O/cb_city_fused.json 2> $O/cb_city_fused.err; echo "cb city fused rc $?"
timeout 300 python bench.py $B $C --obs query --parity-envs 0 > $O/cb_city_query.json 2> $O/cb_city_query.err; echo "cb city query rc $?"
timeout 200 python bench.py $B --steps 200 --warmup 50 > $O/cb_toy.json 2> $O/cb_toy.err
for f in $O/cb_*.json; do python - "$f" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]/1e6,2), "M", round(d["value_min"]/1e6,1), round(d["value_max"]/1e6,1), "ms", round(d["ms_per_step"],4), "parity", (d.get("parity") or {}).get("ok"), (d.get("parity") or {}).get("observation_checks"))
except Exception as e: print(sys.argv[1], "ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-800:])
P
done
