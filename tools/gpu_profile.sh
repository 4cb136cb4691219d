#!/bin/bash
# THE recipe behind profiles/: everything the roofline numbers of `python bench.py` come from, on the GPU box, from ONE build.
#   usage (from the build container):   gpurun -- "GIT_HEAD=$(git rev-parse --short HEAD) bash tools/gpu_profile.sh <tag> [bench flags]"
#   then copy gpurun_out/<tag>/{<tag>_rocprofv3.md,latest_pmc.json,pattern_ceiling.json,bench_line.json} into profiles/.
# Passes (each under its own timeout; one PMC counter set per pass, --kernel-trace only — never combined with other trace domains):
#   1. the un-profiled default bench line                        -> bench_line.json
#   2. tools/hbm_pattern_bench (the step kernel's access pattern, no simulation work) -> pattern_ceiling.json
#   3. rocprofv3 --kernel-trace --stats                          -> trace/
#   4. rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ_* (three runs) -> fetch/ write/ sq/
#   5. tools/refresh_pmc.py: latest_pmc.json (stamped with GIT_HEAD, the step kernels' code-object key + sha, the run's value),
#      the markdown summary, and the per-dispatch rows of every pass as *_dispatches.csv.gz (the raw .db files do not come back).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
tag=$1; shift
O=gpurun_out/$tag
mkdir -p $O
timeout 600 python bench.py "$@" > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc $?"
timeout 120 tools/hbm_pattern_bench --json-out $O/pattern_ceiling.json > $O/pattern_line.json 2> $O/pattern.err; echo "pattern rc $?"
B="python bench.py --steps 60 --warmup 50 --repeats 1 --no-cpu --no-episode --parity-envs 0 $*"
timeout 200 rocprofv3 --kernel-trace --stats -d $O/trace -o r -- python bench.py --steps 200 --warmup 50 --repeats 1 --no-cpu --no-episode --parity-envs 0 "$@" > $O/trace_line.json 2> $O/trace.err; echo "trace rc $?"
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o r -- $B > $O/fetch_line.json 2> $O/fetch.err; echo "fetch rc $?"
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o r -- $B > $O/write_line.json 2> $O/write.err; echo "write rc $?"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/sq -o r -- $B > $O/sq_line.json 2> $O/sq.err; echo "sq rc $?"
python tools/refresh_pmc.py $O profiles/${tag}_rocprofv3.md $O
# ---- citi_bike (CB=1): the toy of BASELINE config 4 and city.800s (the reference's own topology size, sustained: a window that
# spans several decision ticks, bounded steps), same passes, summarised into <tag>_citi_bike.md + latest_pmc_citi_bike.json
if [ "${CB:-0}" = "1" ]; then
  rm -f $O/${tag}_citi_bike.md $O/latest_pmc_citi_bike.json
  cb_pass() {  # <name> <bench flags...>
    name=$1; shift
    C=$O/cb_$name; mkdir -p $C
    timeout 200 python bench.py --scenario citi_bike --no-cpu "$@" > $C/bench_line.json 2> $C/bench_line.err; echo "cb $name bench rc $?"
    timeout 200 rocprofv3 --kernel-trace --stats -d $C/trace -o r -- python bench.py --scenario citi_bike --no-cpu "$@" > $C/trace_line.json 2> $C/trace.err; echo "cb $name trace rc $?"
    timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $C/fetch -o r -- python bench.py --scenario citi_bike --no-cpu "$@" > $C/fetch_line.json 2> $C/fetch.err; echo "cb $name fetch rc $?"
    timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $C/write -o r -- python bench.py --scenario citi_bike --no-cpu "$@" > $C/write_line.json 2> $C/write.err; echo "cb $name write rc $?"
    python tools/refresh_pmc_citi_bike.py $C $name $O
  }
  cb_pass toy3s --bounded-budget 0
  cb_pass city800 --topology city.800s --envs 4096 --durations 2880 --steps 900 --warmup 300 --bounded-budget 0 --step-budget 64 --specialize 1
fi
find $O -name "*.db" -delete; find $O -type d -empty -delete; du -sh $O
