#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${1:-r02b}
mkdir -p $O
for m in ${2:-3}; do
  timeout 300 python tools/phase_profile.py --specialized --step-mode $m > $O/phase_m$m.txt 2>&1
  cat $O/phase_m$m.txt | tail -17
done
