#!/bin/bash
O=gpurun_out/r05o; mkdir -p $O
export PYTHONPATH=$PWD
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --scenario citi_bike --no-cpu --steps 400 --warmup 100 --repeats 5 --parity-envs 0 > $O/cb_${tag}.json 2> $O/cb_${tag}.err; }
for rep in 1 2; do
  run base_r$rep MARO_AMD_LIB=$PWD/variants/base/libmaro_amd.so MARO_AMD_CSRC=$PWD/variants/csrc_base MARO_AMD_SPEC_CACHE=$PWD/variants/cache_base
  run new_r$rep X=1
done
timeout 300 python bench.py --scenario citi_bike --no-cpu --envs 32768 --steps 400 --warmup 100 --repeats 3 --parity-envs 0 > $O/cb_new_32768.json 2> $O/cb_new_32768.err
MRX_CB_LANES=16 timeout 300 python bench.py --scenario citi_bike --no-cpu --steps 100 --warmup 20 --repeats 2 --parity-envs 4 > $O/cb_new_lanes16.json 2> $O/cb_new_lanes16.err
timeout 900 python -m pytest tests/test_gpu_citi_bike.py tests/test_gpu_citi_bike_api.py tests/test_gpu_specialized.py -m gpu -x -q > $O/pytest_cb.log 2>&1; echo "pytest rc $?" >> $O/pytest_cb.log
python - <<P
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d["value"]/1e6,2), "M", round(d["ms_per_step"]*1e3,1), "us", "min/max", round(d.get("value_min",0)/1e6,1), round(d.get("value_max",0)/1e6,1), "parity", (d.get("parity") or {}).get("ok"), d["config"].get("specialized_kernels"))
    except Exception as e: print(f, "FAILED", e, open(f.replace('.json','.err')).read()[-300:])
P
tail -3 $O/pytest_cb.log
