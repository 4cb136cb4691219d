cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O; shift
CF="--policy dqn --collect --ring 8 --envs 8192 --no-cpu --parity-envs 0 --steps 64 --warmup 16 --repeats 2 $*"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o r -- python bench.py $CF > $O/trace_line.json 2> $O/trace.err; echo "trace rc $?"
python tools/rocprof_summary.py $(find $O/trace -name "*_results.db" | head -1) 2>&1 | grep -E "mrx_k_cim_(dqn|step)" | cut -c1-120
rm -rf $O/trace
