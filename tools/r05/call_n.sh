#!/bin/bash
O=gpurun_out/r05n; mkdir -p $O
export PYTHONPATH=$PWD
for rep in 1 2; do
for a in 0 1; do
  MRX_CIM_SCHED_APPEND=$a timeout 300 python bench.py --steps 300 --warmup 50 --repeats 3 --no-cpu --secondary 0 --parity-envs 16 --no-episode > $O/head_a${a}_r$rep.json 2> $O/head_a${a}_r$rep.err
  MRX_CIM_SCHED_APPEND=$a timeout 300 python bench.py --policy dqn --collect --envs 8192 --ring 8 --steps 64 --warmup 16 --repeats 3 --groups 2 --no-cpu --parity-envs 6 > $O/collect_a${a}_r$rep.json 2> $O/collect_a${a}_r$rep.err
done
done
timeout 600 python -m pytest tests/test_gpu_step_modes.py tests/test_gpu_bench_parity.py tests/test_gpu_dqn.py tests/test_sampler.py tests/test_gpu_vector_env.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
python - <<P
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d["value"]/1e6,2), "M", round(d["ms_per_step"]*1e3,1), "us", "min/max", round(d.get("value_min",0)/1e6,1), round(d.get("value_max",0)/1e6,1), "parity", (d.get("parity") or {}).get("ok"))
    except Exception as e: print(f, "FAILED", e, open(f.replace('.json','.err')).read()[-400:])
P
tail -3 $O/pytest.log
