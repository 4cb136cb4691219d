#!/bin/bash
# usage: gpu_quick.sh <tag> [pytest-target|-] <bench variant>...   (each variant = a quoted string of bench.py flags)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
tag=$1; shift
O=gpurun_out/$tag
mkdir -p $O
t=$1; shift
if [ "$t" != "-" ]; then
  timeout 1200 python -m pytest $t -x -q > $O/pytest.log 2>&1
  echo "pytest rc $?" >> $O/pytest.log
  tail -4 $O/pytest.log
fi
i=0
for v in "$@"; do
  i=$((i+1))
  timeout 300 python bench.py --no-cpu $v > $O/bench_$i.json 2> $O/bench_$i.err
  echo "[$v] rc $? $(python -c "import json; d=json.load(open('$O/bench_$i.json')); print(round(d['value']/1e6,1), 'M', round(d['ms_per_step'],4), 'ms mode', d['config']['step_mode'], 'kernel_ms', round(d['roofline']['kernel_ms'],4))" 2>&1 | tail -1)"
done
