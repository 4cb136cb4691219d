#!/bin/bash
# round 4, call D: citi_bike fused observation, the driver's line with the device-resident collection loop
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r04d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_citi_bike.py tests/test_gpu_citi_bike_api.py tests/test_gpu_dqn.py tests/test_sampler.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
B="--scenario citi_bike --no-cpu --bounded-budget 0 --repeats 3 --steps 200 --warmup 50"
timeout 200 python bench.py $B > $O/cb_toy_fused.json 2> $O/cb_toy_fused.err; echo "cb fused rc $?"
timeout 200 python bench.py $B --obs query > $O/cb_toy_query.json 2> $O/cb_toy_query.err; echo "cb query rc $?"
timeout 200 python bench.py $B --envs 32768 > $O/cb_toy_32768.json 2> $O/cb_toy_32768.err
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc $?"
for f in $O/cb_*.json; do python - "$f" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]/1e6,2), "M", round(d["value_min"]/1e6,1), round(d["value_max"]/1e6,1), "ms", round(d["ms_per_step"],4), "parity", (d.get("parity") or {}).get("ok"), (d.get("parity") or {}).get("observation_checks"))
except Exception as e: print(sys.argv[1], "ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-600:])
P
done
python - <<'P'
import json
d=json.loads(open('gpurun_out/r04d/bench_driver.json').read().strip().splitlines()[-1])
print('headline', d['value']/1e6, d['value_end_to_end']/1e6, d['parity']['ok'], d['roofline']['frac'])
for k,v in d['secondary'].items(): print(k, v['value']/1e6, v['ms_per_step'], v['parity']['ok'], v['roofline']['frac'])
P
