#!/bin/bash
# kernel trace of a short bench run, summarised on the box. usage: gpu_trace.sh <tag> [bench flags]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
tag=$1; shift
O=gpurun_out/$tag
mkdir -p $O
timeout 200 rocprofv3 --kernel-trace --stats -d $O/trace -o r -- python bench.py --steps 100 --warmup 50 --no-cpu --no-episode --parity-envs 0 "$@" > $O/trace_line.json 2> $O/trace.err; echo "trace rc $?"
python tools/rocprof_summary.py $(find $O/trace -name "r_results.db") 2>&1 | grep "mrx_k\|kernel |" | cut -c1-200 | tee $O/trace_summary.md
find $O -name "*.db" -delete
