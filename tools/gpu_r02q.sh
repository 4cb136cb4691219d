#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for f in "" "-DMRX_NO_OPT_C" "-DMRX_NO_OPT_B" "-DMRX_NO_OPT_B -DMRX_NO_OPT_C"; do
  MARO_AMD_SPEC_FLAGS="$f" timeout 200 python bench.py --no-cpu --steps 300 --warmup 100 --no-episode --parity-envs 0 --step-mode 2 --groups 3 > /tmp/b.json 2>/tmp/b.err
  echo "[$f] $(python -c "import json; d=json.load(open('/tmp/b.json')); print(round(d['value']/1e6,1), 'M', d['config']['specialized_kernels'])" 2>&1 | tail -1)"
done
