#!/bin/bash
# citi_bike: envs-per-wave sweep (MRX_CB_LANES).  usage: gpu_cb_lanes.sh <tag> [test]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${1:-cblanes}
mkdir -p $O
if [ "$2" = "test" ]; then timeout 900 python -m pytest tests/test_gpu_citi_bike.py tests/test_gpu_specialized.py -x -q -k "citi or cb" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log; fi
run() {  # lanes, bench flags
  l=$1; shift
  f=$O/b_$(echo "$l $*" | tr -d ' -' | tr '.' '_')
  MRX_CB_LANES=$l timeout 200 python bench.py --scenario citi_bike $* --no-cpu --steps 300 --warmup 50 > $f.json 2> $f.err
  echo "lanes $l [$*]: $(python -c "import json; d=json.load(open('$f.json')); print(round(d['value']/1e6,3), 'M', 'kernel_ms', round(d['roofline']['kernel_ms'],4))" 2>&1 | tail -1)"
}
for l in ${LANES_A:-0 4 16 64}; do run $l --envs 4096; done
for l in ${LANES_B:-0 16 64}; do run $l --envs 32768; done
for l in ${LANES_C:-0 1 4 8}; do run $l --envs 4096 --topology city.180s; done
