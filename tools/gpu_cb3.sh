#!/bin/bash
# citi_bike: tests, bench lines (lockstep + bounded leg).  usage: gpu_cb3.sh <tag> [test] ; env BUDGETS_TOY / BUDGETS_CITY
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${1:-cb3}
mkdir -p $O
if [ "$2" = "test" ]; then timeout 900 python -m pytest tests/test_gpu_citi_bike.py tests/test_gpu_specialized.py -x -q -k "citi or cb or bounded" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log; fi
run() {
  f=$O/b_$(echo "$*" | tr -d ' -' | tr '.' '_')
  timeout 200 python bench.py --scenario citi_bike $* --no-cpu --steps 300 --warmup 50 > $f.json 2> $f.err
  echo "[$*]: $(python -c "import json; d=json.load(open('$f.json')); b=d.get('bounded_steps') or {}; print(round(d['value']/1e6,3), 'M lockstep', round(d['ms_per_step'],4), 'ms kernel', round(d['roofline']['kernel_ms'],4), '| bounded', b.get('budget_records'), round(b.get('value',0)/1e6,3), 'M', round(b.get('ms_per_call',0),4), 'ms', round(b.get('decisions_per_call_per_env',0),3))" 2>&1 | tail -1)"
}
for b in ${BUDGETS_TOY:-24}; do run --envs 4096 --bounded-budget $b; done
for b in ${BUDGETS_TOY32:-32}; do run --envs 32768 --bounded-budget $b; done
for b in ${BUDGETS_CITY:-8 16}; do run --envs 4096 --topology city.180s --bounded-budget $b; done
