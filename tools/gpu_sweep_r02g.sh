#!/bin/bash
# groups x batch-size sweep of the default launch form (specialised kernels; bench plans other than 16384/3 compile on the box)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${1:-r02g}
mkdir -p $O
run() {  # env-prefix, flags
  pre=$1; shift
  f=$O/b_$(echo "$pre $*" | tr -d ' -=' | tr '.' '_')
  env $pre timeout 240 python bench.py --no-cpu --no-episode --parity-envs 0 --steps 300 --warmup 100 $* > $f.json 2> $f.err
  echo "[$pre $*] $(python -c "import json; d=json.load(open('$f.json')); print(round(d['value']/1e6,1), 'M', round(d['ms_per_step'],4), 'ms kernel', round(d['roofline']['kernel_ms'],4), 'inflight', round(d['roofline']['launches_in_flight'],2), 'spec', d['config']['specialized_kernels'])" 2>&1 | tail -1)"
}
run A=1 --groups 2
run A=1 --groups 3
run A=1 --groups 4
run GPU_MAX_HW_QUEUES=8 --groups 4
run GPU_MAX_HW_QUEUES=8 --groups 6
run A=1 --groups 3 --envs 32768
run GPU_MAX_HW_QUEUES=8 --groups 6 --envs 32768
run A=1 --groups 3 --envs 8192
