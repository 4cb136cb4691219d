#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${1:-cb}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_citi_bike.py tests/test_gpu_specialized.py -x -q -k "citi or cb" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
for n in 4096 32768; do
  timeout 200 python bench.py --scenario citi_bike --envs $n --no-cpu --steps 300 --warmup 50 > $O/bench_$n.json 2> $O/bench_$n.err
  echo "citi_bike $n envs: $(python -c "import json; d=json.load(open('$O/bench_$n.json')); print(round(d['value']/1e6,1), 'M', round(d['ms_per_step'],4), 'ms spec', d['config']['specialized_kernels'], round(d['roofline']['kernel_ms'],4))" 2>&1 | tail -1)"
done
