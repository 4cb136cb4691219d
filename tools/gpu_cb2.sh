#!/bin/bash
# citi_bike bench lines: toy.3s_4t at 4096 / 32768 envs and city.180s at 4096 envs (+ optional pytest)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${1:-cb2}
mkdir -p $O
if [ "$2" = "test" ]; then timeout 900 python -m pytest tests/test_gpu_citi_bike.py tests/test_gpu_specialized.py -x -q -k "citi or cb" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log; fi
for v in "--envs 4096" "--envs 32768" "--envs 4096 --topology city.180s"; do
  f=$O/bench_cb_$(echo $v | tr -d ' -' | tr '.' '_')
  timeout 300 python bench.py --scenario citi_bike $v --no-cpu --steps 300 --warmup 50 > $f.json 2> $f.err
  echo "citi_bike $v: rc $? $(python -c "import json; d=json.load(open('$f.json')); print(round(d['value']/1e6,3), 'M', round(d['ms_per_step'],4), 'ms spec', d['config']['specialized_kernels'], 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'tbar', round(d['config']['mean_ticks_per_env_step'],3))" 2>&1 | tail -1)"
done
