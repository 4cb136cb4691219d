#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for f in "-DMRX_EXP_STORE=3" "-DMRX_EXP_STORE=4"; do
  echo "== $f"
  MARO_AMD_SPEC_FLAGS="$f" timeout 300 python tools/phase_profile.py --specialized --step-mode 3 2>&1 | grep "raw\|mean"
done
