#!/bin/bash
O=gpurun_out/r05f; mkdir -p $O
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_citi_bike.py tests/test_gpu_citi_bike_api.py -m gpu -x -q > $O/pytest_cb.log 2>&1; echo "pytest rc $?" >> $O/pytest_cb.log
for rep in 1 2; do
  timeout 300 python bench.py --scenario citi_bike --no-cpu --steps 400 --warmup 100 --repeats 5 > $O/cb_toy_r$rep.json 2> $O/cb_toy_r$rep.err
done
timeout 300 python bench.py --scenario citi_bike --no-cpu --steps 20 --warmup 5 > $O/cb_toy_k20.json 2> $O/cb_toy_k20.err
timeout 300 python bench.py --scenario citi_bike --no-cpu --envs 32768 --steps 400 --warmup 100 --repeats 3 --parity-envs 0 > $O/cb_toy_32768.json 2> $O/cb_toy_32768.err
timeout 400 python bench.py --scenario citi_bike --no-cpu --topology city.800s --envs 4096 --durations 2880 --steps 900 --warmup 300 --repeats 3 --bounded-budget 0 --step-budget 64 --specialize 1 > $O/cb_city800.json 2> $O/cb_city800.err
python - <<P
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d["value"]/1e6,2), "M", round(d["ms_per_step"]*1e3,1), "us", "min/max", round(d.get("value_min",0)/1e6,1), round(d.get("value_max",0)/1e6,1), "parity", (d.get("parity") or {}).get("ok"))
    except Exception as e: print(f, "FAILED", e)
P
tail -4 $O/pytest_cb.log
