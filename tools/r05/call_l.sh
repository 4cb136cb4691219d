#!/bin/bash
O=gpurun_out/r05l; mkdir -p $O
export PYTHONPATH=$PWD
for m in 0 1 4; do
  timeout 300 python bench.py --steps 300 --warmup 50 --repeats 3 --no-cpu --secondary 0 --parity-envs 0 --no-episode --step-mode $m > $O/mode$m.json 2> $O/mode$m.err
done
for g in 2 4; do
  timeout 300 python bench.py --steps 300 --warmup 50 --repeats 3 --no-cpu --secondary 0 --parity-envs 0 --no-episode --groups $g > $O/groups$g.json 2> $O/groups$g.err
done
python - <<P
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d["value"]/1e6,2), "M", round(d["ms_per_step"]*1e3,1), "us", "min/max", round(d.get("value_min",0)/1e6,1), round(d.get("value_max",0)/1e6,1), d["config"].get("step_mode"))
    except Exception as e: print(f, "FAILED", e, open(f.replace('.json','.err')).read()[-300:])
P
