#!/bin/bash
O=gpurun_out/r05d; mkdir -p $O
export PYTHONPATH=$PWD
rocm-smi --showperflevel --showclocks --showpower > $O/smi_idle.txt 2>&1
rocm-smi --showclkfrq > $O/smi_clkfrq.txt 2>&1
# headline loop, long window, clocks sampled while it runs
(MARO_AMD_LIB=$PWD/variants/new2/libmaro_amd.so python bench.py --steps 6000 --warmup 100 --repeats 3 --no-cpu --secondary 0 --parity-envs 0 --no-episode > $O/headline_long.json 2> $O/headline_long.err) &
BP=$!
sleep 12
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower --showuse 2>&1 | grep -E "sclk|mclk|fclk|socclk|Power|busy" >> $O/smi_headline.txt; echo --- >> $O/smi_headline.txt; sleep 1; done
wait $BP
# collect loop, long
(MARO_AMD_LIB=$PWD/variants/new2/libmaro_amd.so python bench.py --policy dqn --collect --envs 8192 --ring 8 --steps 2000 --warmup 16 --repeats 3 --groups 2 --no-cpu --parity-envs 0 > $O/collect_long.json 2> $O/collect_long.err) &
BP=$!
sleep 12
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower --showuse 2>&1 | grep -E "sclk|mclk|fclk|socclk|Power|busy" >> $O/smi_collect.txt; echo --- >> $O/smi_collect.txt; sleep 1; done
wait $BP
# citi_bike, long
(python bench.py --scenario citi_bike --steps 20000 --warmup 100 --repeats 3 --no-cpu --parity-envs 0 > $O/cb_long.json 2> $O/cb_long.err) &
BP=$!
sleep 10
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower --showuse 2>&1 | grep -E "sclk|mclk|fclk|socclk|Power|busy" >> $O/smi_cb.txt; echo --- >> $O/smi_cb.txt; sleep 1; done
wait $BP
python - <<P
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d["value"]/1e6,2), "M", round(d["ms_per_step"]*1e3,1), "us")
    except Exception as e: print(f, "FAILED", e)
P
cat $O/smi_idle.txt | head -40; echo; head -30 $O/smi_headline.txt; echo; head -30 $O/smi_collect.txt; echo; head -30 $O/smi_cb.txt; cat $O/smi_clkfrq.txt | head -60
