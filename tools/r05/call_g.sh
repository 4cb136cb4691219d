#!/bin/bash
O=gpurun_out/r05h; mkdir -p $O
export PYTHONPATH=$PWD
BASE="MARO_AMD_CSRC=$PWD/variants/csrc_base MARO_AMD_SPEC_CACHE=$PWD/variants/cache_base"
for rep in 1 2 3; do
  env MARO_AMD_CSRC=$PWD/variants/csrc_base MARO_AMD_SPEC_CACHE=$PWD/variants/cache_base timeout 300 python bench.py --scenario citi_bike --no-cpu --steps 400 --warmup 100 --repeats 5 --parity-envs 0 > $O/cb_base_r$rep.json 2> $O/cb_base_r$rep.err
  timeout 300 python bench.py --scenario citi_bike --no-cpu --steps 400 --warmup 100 --repeats 5 --parity-envs 0 > $O/cb_new_r$rep.json 2> $O/cb_new_r$rep.err
done


python - <<P
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d["value"]/1e6,2), "M", round(d["ms_per_step"]*1e3,1), "us", "min/max", round(d.get("value_min",0)/1e6,1), round(d.get("value_max",0)/1e6,1), d["config"].get("code_object_key"))
    except Exception as e: print(f, "FAILED", e)
P
echo BASE; grep -v amdgpu.ids $O/phases_base.txt | tail -22; echo NEW; grep -v amdgpu.ids $O/phases_new.txt | tail -22
