#!/bin/bash
# round 4, call G: the driver's N > 1 command on two gloo ranks (one GPU), then the driver's N = 1 command
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r04g; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_bench_contract.py -m gpu -x -q -k "two_ranks or default_line" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -30 $O/pytest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc $?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/r04g/bench_driver.json').read().strip().splitlines()[-1])
print('headline', d['value']/1e6, d['value_end_to_end']/1e6, d['parity']['ok'], d['roofline']['frac'], d['gpu_seconds_total'])
for k,v in d['secondary'].items(): print(k, v['value']/1e6, v['ms_per_step'], v['parity']['ok'], v['roofline']['frac'], v['roofline']['basis'][:60])
P
