#!/bin/bash
O=gpurun_out/r05m; mkdir -p $O
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_order_fast.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
for rep in 1 2; do
  timeout 600 python bench.py --steps 200 --warmup 20 --repeats 3 --no-cpu --secondary 0 --parity-envs 16 > $O/headline_r$rep.json 2> $O/headline_r$rep.err
done
timeout 600 python tools/gpu_fuzz_sweep.py 40000 60 > $O/fuzz.log 2>&1; echo "fuzz rc $?" >> $O/fuzz.log
python - <<P
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d["value"]/1e6,2), "M", round(d["ms_per_step"]*1e3,1), "us", "e2e", round(d.get("value_end_to_end",0)/1e6,1), "reset_ms", d.get("config",{}).get("reset_ms_whole_batch"), "parity", (d.get("parity") or {}).get("ok"))
    except Exception as e: print(f, "FAILED", e)
P
tail -3 $O/pytest.log; tail -3 $O/fuzz.log
