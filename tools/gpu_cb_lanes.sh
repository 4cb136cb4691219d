#!/bin/bash
# citi_bike: envs-per-wave sweep (MRX_CB_LANES).  usage: gpu_cb_lanes.sh <tag>
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${1:-cblanes}
mkdir -p $O
run() {  # lanes, bench flags
  l=$1; shift
  f=$O/b_$(echo "$l $*" | tr -d ' -' | tr '.' '_')
  MRX_CB_LANES=$l timeout 200 python bench.py --scenario citi_bike $* --no-cpu --steps 300 --warmup 50 > $f.json 2> $f.err
  echo "lanes $l [$*]: $(python -c "import json; d=json.load(open('$f.json')); print(round(d['value']/1e6,3), 'M', 'kernel_ms', round(d['roofline']['kernel_ms'],4))" 2>&1 | tail -1)"
}
for l in 1 2 4 8 64; do run $l --envs 4096; done
for l in 2 4 8 16 64; do run $l --envs 32768; done
for l in 1 4 64; do run $l --envs 4096 --topology city.180s; done
