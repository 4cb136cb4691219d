#!/bin/bash
O=gpurun_out/r05e; mkdir -p $O
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_order_fast.py -m gpu -x -q > $O/pytest_order_fast.log 2>&1; echo "pytest rc $?" >> $O/pytest_order_fast.log
timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_specialized.py tests/test_gpu_dqn.py tests/test_sampler.py -m gpu -x -q > $O/pytest_golden.log 2>&1; echo "pytest rc $?" >> $O/pytest_golden.log
for rep in 1 2; do
for f in 0 1; do
  MRX_ORDER_FAST=$f timeout 600 python bench.py --steps 200 --warmup 20 --repeats 3 --no-cpu --secondary 0 --parity-envs 16 > $O/headline_f${f}_r$rep.json 2> $O/headline_f${f}_r$rep.err
done
done
# clocks while the headline loop runs (long window)
(python bench.py --steps 3000 --warmup 100 --repeats 3 --no-cpu --secondary 0 --parity-envs 0 --no-episode > $O/headline_long.json 2> $O/headline_long.err) &
BP=$!
sleep 9
for i in 1 2 3 4 5 6 7 8; do rocm-smi --showclocks --showpower --showuse 2>&1 | grep -E "sclk|Power|busy" | tr '\n' ' ' >> $O/smi_headline.txt; echo >> $O/smi_headline.txt; sleep 0.5; done
wait $BP
(python bench.py --policy dqn --collect --envs 8192 --ring 8 --steps 1500 --warmup 16 --repeats 3 --groups 2 --no-cpu --parity-envs 0 > $O/collect_long.json 2> $O/collect_long.err) &
BP=$!
sleep 9
for i in 1 2 3 4 5 6 7 8; do rocm-smi --showclocks --showpower --showuse 2>&1 | grep -E "sclk|Power|busy" | tr '\n' ' ' >> $O/smi_collect.txt; echo >> $O/smi_collect.txt; sleep 0.5; done
wait $BP
(python bench.py --scenario citi_bike --steps 1500 --warmup 100 --repeats 6 --no-cpu --parity-envs 0 > $O/cb_long.json 2> $O/cb_long.err) &
BP=$!
sleep 8
for i in 1 2 3 4 5 6 7 8; do rocm-smi --showclocks --showpower --showuse 2>&1 | grep -E "sclk|Power|busy" | tr '\n' ' ' >> $O/smi_cb.txt; echo >> $O/smi_cb.txt; sleep 0.5; done
wait $BP
python - <<P
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d["value"]/1e6,2), "M", round(d["ms_per_step"]*1e3,1), "us", "e2e", round(d.get("value_end_to_end",0)/1e6,1), "reset_ms", d.get("config",{}).get("reset_ms_whole_batch"), "parity", (d.get("parity") or {}).get("ok"))
    except Exception as e: print(f, "FAILED", e)
P
tail -4 $O/pytest_order_fast.log; tail -4 $O/pytest_golden.log; echo; cat $O/smi_headline.txt; echo; cat $O/smi_collect.txt; echo; cat $O/smi_cb.txt
