#!/bin/bash
O=gpurun_out/r05j; mkdir -p $O
export PYTHONPATH=$PWD
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout 300 python bench.py --scenario citi_bike --no-cpu --steps 400 --warmup 100 --repeats 5 --parity-envs 0 > $O/cb_${tag}.json 2> $O/cb_${tag}.err
}
for rep in 1 2; do
  run base_r$rep MARO_AMD_LIB=$PWD/variants/base/libmaro_amd.so MARO_AMD_CSRC=$PWD/variants/csrc_base MARO_AMD_SPEC_CACHE=$PWD/variants/cache_base
  run new_r$rep X=1
done
timeout 400 python bench.py --scenario citi_bike --no-cpu --topology city.800s --envs 4096 --durations 2880 --steps 900 --warmup 300 --repeats 3 --bounded-budget 0 --step-budget 64 --specialize 1 > $O/city800_new.json 2> $O/city800_new.err
env MARO_AMD_LIB=$PWD/variants/base/libmaro_amd.so MARO_AMD_CSRC=$PWD/variants/csrc_base MARO_AMD_SPEC_CACHE=$PWD/variants/cache_base timeout 400 python bench.py --scenario citi_bike --no-cpu --topology city.800s --envs 4096 --durations 2880 --steps 900 --warmup 300 --repeats 3 --bounded-budget 0 --step-budget 64 --specialize 1 > $O/city800_base.json 2> $O/city800_base.err
timeout 900 python -m pytest tests/test_gpu_citi_bike.py tests/test_gpu_citi_bike_api.py -m gpu -x -q > $O/pytest_cb.log 2>&1; echo "pytest rc $?" >> $O/pytest_cb.log
python - <<P
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d["value"]/1e6,2), "M", round(d["ms_per_step"]*1e3,1), "us", "min/max", round(d.get("value_min",0)/1e6,1), round(d.get("value_max",0)/1e6,1), "parity", (d.get("parity") or {}).get("ok"))
    except Exception as e: print(f, "FAILED", e)
P
tail -3 $O/pytest_cb.log
