#!/bin/bash
# kernel trace of the default bench (incl. reset + the end-to-end episode) -> gpurun_out/<tag>/trace_summary.md
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${1:-trace_final}
mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace -o r -- python bench.py --no-cpu --parity-envs 0 --steps 300 --warmup 100 > $O/trace_line.json 2> $O/trace.err; echo "trace rc $?"
db=$(find $O/trace -name "r_results.db" | head -1)
python tools/rocprof_summary.py $db > $O/trace_summary.md 2>&1
head -14 $O/trace_summary.md
find $O -name "*.db" -delete
