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
if [ "${HEADLINE:-1}" = "1" ]; then   # (HEADLINE=0: only the optional blocks below, e.g. CB=1 after a change to the citi_bike kernels alone)
timeout 900 python bench.py "$@" > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc $?"
[ -x tools/hbm_pattern_bench ] || hipcc -O3 --offload-arch=gfx950 -o tools/hbm_pattern_bench tools/hbm_pattern_bench.hip   # (built from source: the binary is not tracked)
timeout 120 tools/hbm_pattern_bench --json-out $O/pattern_ceiling.json > $O/pattern_line.json 2> $O/pattern.err; echo "pattern rc $?"
B="python bench.py --steps 60 --warmup 50 --repeats 1 --no-cpu --no-episode --parity-envs 0 --secondary 0 $*"
timeout 200 rocprofv3 --kernel-trace --stats -d $O/trace -o r -- python bench.py --steps 200 --warmup 50 --repeats 1 --no-cpu --no-episode --parity-envs 0 --secondary 0 "$@" > $O/trace_line.json 2> $O/trace.err; echo "trace rc $?"
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o r -- $B > $O/fetch_line.json 2> $O/fetch.err; echo "fetch rc $?"
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o r -- $B > $O/write_line.json 2> $O/write.err; echo "write rc $?"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/sq -o r -- $B > $O/sq_line.json 2> $O/sq.err; echo "sq rc $?"
python tools/refresh_pmc.py $O profiles/${tag}_rocprofv3.md $O
fi
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
  cb_pass toy3s_32768 --envs 32768 --bounded-budget 0
  cb_pass city800 --topology city.800s --envs 4096 --durations 2880 --steps 900 --warmup 300 --bounded-budget 0 --step-budget 96 --replay-period 4 --cb-groups 2 --specialize 1
fi
# ---- config 5 (COLLECT=1): the DQN collection loop at 8192 envs per GPU — kernel trace + the two PMC passes over every kernel of the loop
if [ "${COLLECT:-0}" = "1" ]; then
  rm -f $O/${tag}_collect.md $O/latest_pmc_collect.json
  for envs in ${COLLECT_ENVS:-8192}; do
    C=$O/collect_$envs; mkdir -p $C
    CF="--policy dqn --collect --ring 8 --envs $envs --groups 2 --no-cpu --parity-envs 0"
    timeout 300 python bench.py $CF --steps 64 --warmup 16 > $C/bench_line.json 2> $C/bench_line.err; echo "collect $envs bench rc $?"
    timeout 300 rocprofv3 --kernel-trace --stats -d $C/trace -o r -- python bench.py $CF --steps 64 --warmup 16 --repeats 2 > $C/trace_line.json 2> $C/trace.err; echo "collect $envs trace rc $?"
    timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $C/fetch -o r -- python bench.py $CF --steps 64 --warmup 16 --repeats 1 > $C/fetch_line.json 2> $C/fetch.err; echo "collect $envs fetch rc $?"
    timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $C/write -o r -- python bench.py $CF --steps 64 --warmup 16 --repeats 1 > $C/write_line.json 2> $C/write.err; echo "collect $envs write rc $?"
    python tools/refresh_pmc_collect.py $C $envs $O
  done
fi
# ---- BIG=1: the headline's PMC passes at 65536 envs per GPU (a working set 4x the Infinity Cache: are the bytes per env-step the same?)
if [ "${BIG:-0}" = "1" ]; then
  Bg=$O/big65536; mkdir -p $Bg
  BB="python bench.py --envs 65536 --steps 60 --warmup 50 --repeats 1 --no-cpu --no-episode --parity-envs 0"
  timeout 300 python bench.py --envs 65536 --steps 200 --warmup 50 --repeats 3 --no-cpu --no-episode --parity-envs 0 > $Bg/bench_line.json 2> $Bg/bench_line.err; echo "big bench rc $?"
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $Bg/fetch -o r -- $BB > $Bg/fetch_line.json 2> $Bg/fetch.err; echo "big fetch rc $?"
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $Bg/write -o r -- $BB > $Bg/write_line.json 2> $Bg/write.err; echo "big write rc $?"
  mkdir -p $Bg/out; python tools/refresh_pmc.py $Bg profiles/${tag}_65536_rocprofv3.md $Bg/out; cp $Bg/out/latest_pmc.json $O/pmc_65536.json; cp $Bg/out/${tag}_65536_rocprofv3.md $O/ 2>/dev/null
fi
# ---- ISA=1: what the ISA budget (profiles/<tag>_isa_budget.md) needs of the headline step kernel: wave cycles per phase of the plan-specialised
# kernel (a code object built with -DMRX_PROFILE_PHASES) and a second set of SQ counters
if [ "${ISA:-0}" = "1" ]; then
  timeout 300 python tools/phase_profile.py --specialized --envs 16384 > $O/phase_profile_spec.txt 2> $O/phase_profile_spec.err; echo "phase profile rc $?"
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU -d $O/sq2 -o r -- $B > $O/sq2_line.json 2> $O/sq2.err; echo "sq2 rc $?"
  python tools/rocprof_summary.py $(find $O/sq2 -name "*_results.db" | head -1) 2> /dev/null | grep -E "kernel|mrx_k_cim" | cut -c1-600 > $O/sq2_summary.md
fi
find $O -name "*.db" -delete; find $O -type d -empty -delete; du -sh $O
