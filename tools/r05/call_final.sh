#!/bin/bash
# the round's final GPU call on the committed tree 748ff8a: full GPU suite, smoke, the profiles/ recipe, the driver's command
export PYTHONPATH=$PWD TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/ -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
GIT_HEAD=748ff8a CB=1 COLLECT=1 BIG=1 bash tools/gpu_profile.sh r05 > $O/profile_recipe.log 2>&1
cp $O/latest_pmc.json $O/latest_pmc_citi_bike.json $O/latest_pmc_collect.json profiles/ 2>/dev/null
cp $O/pattern_ceiling.json profiles/ 2>/dev/null
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_command.json 2> $O/driver_command.err; echo "driver rc $?"
timeout 600 python tools/gpu_fuzz_sweep.py 50000 150 > $O/fuzz_generic.log 2>&1; tail -1 $O/fuzz_generic.log
MARO_AMD_SPECIALIZE=1 timeout 600 python tools/gpu_fuzz_sweep.py 51000 30 > $O/fuzz_spec.log 2>&1; tail -1 $O/fuzz_spec.log
tail -3 $O/pytest_gpu.log; tail -3 $O/smoke.log; tail -25 $O/profile_recipe.log
python - <<P
import json
d=json.loads(open("$O/driver_command.json").read().strip().splitlines()[-1])
print("headline", round(d["value"]/1e6,1), "e2e", round(d.get("value_end_to_end",0)/1e6,1), "frac", d["roofline"]["frac"], "parity", d["parity"]["ok"])
for k,v in d.get("secondary",{}).items(): print(k, round(v["value"]/1e6,1), "frac", (v.get("roofline") or {}).get("frac"), "parity", (v.get("parity") or {}).get("ok"))
print("ref", d.get("cpu_baseline_reference",{}).get("value"), d.get("cpu_baseline",{}).get("value"))
P
