#!/bin/bash
# usage: gpu_flags.sh "<flags1>" "<flags2>" ...   : bench (sorted launch, 3 groups) with each set of extra spec-build flags
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for f in "$@"; do
  MARO_AMD_SPEC_FLAGS="$f" timeout 200 python bench.py --no-cpu --steps 300 --warmup 100 --no-episode --parity-envs 0 > /tmp/b.json 2>/tmp/b.err
  echo "[$f] $(python -c "import json; d=json.load(open('/tmp/b.json')); print(round(d['value']/1e6,1), 'M', d['config']['specialized_kernels'])" 2>&1 | tail -1)"
done
