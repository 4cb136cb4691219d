#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r04l; mkdir -p $O
timeout 300 python tools/gpu_fuzz_sweep.py 9090 15 > $O/fuzz.log 2>&1; tail -n 2 $O/fuzz.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc $?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/r04l/bench_driver.json').read().strip().splitlines()[-1])
print('headline', d['value']/1e6, d['value_end_to_end']/1e6, d['parity']['ok'], d['roofline']['frac'], d['gpu_seconds_total'])
for k,v in d['secondary'].items(): print(k, v['value']/1e6, v['ms_per_step'], v['parity']['ok'], v['roofline']['frac'], v['roofline']['basis'][:60])
P
