#!/bin/bash
# round 4, call F: the whole GPU suite, the chain probe, and THE profile recipe (headline + citi_bike + collection loop + 65536 envs)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r04f; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
timeout 300 python tools/chain_probe.py > $O/chain_probe.json 2> $O/chain_probe.err; echo "chain rc $?"; cat $O/chain_probe.json
CB=1 COLLECT=1 BIG=1 bash tools/gpu_profile.sh r04 --cpu-seconds 10
