#!/bin/bash
# round 4, call K (final): profile recipe on the final sources, fuzz sweep, the whole GPU suite, the driver's command
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r04k; mkdir -p $O
CB=1 COLLECT=1 bash tools/gpu_profile.sh r04 --cpu-seconds 10 > $O/profile.log 2>&1; tail -3 $O/profile.log | cut -c1-300
timeout 600 python tools/gpu_fuzz_sweep.py 9000 100 > $O/fuzz_generic.log 2>&1; tail -2 $O/fuzz_generic.log
MARO_AMD_SPECIALIZE=1 timeout 900 python tools/gpu_fuzz_sweep.py 9500 25 > $O/fuzz_spec.log 2>&1; tail -2 $O/fuzz_spec.log
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc $?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/r04k/bench_driver.json').read().strip().splitlines()[-1])
print('headline', d['value']/1e6, d['value_end_to_end']/1e6, d['parity']['ok'], d['roofline']['frac'], d['gpu_seconds_total'])
for k,v in d['secondary'].items(): print(k, v['value']/1e6, v['ms_per_step'], v['parity']['ok'], v['roofline']['frac'], v['roofline']['basis'][:60])
P
