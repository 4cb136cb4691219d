#!/bin/bash
# rocprofv3 evidence of `python bench.py` on the GPU box: un-profiled line, kernel trace, FETCH_SIZE / WRITE_SIZE / SQ passes
# (one counter set per pass, --kernel-trace only, every call under a short timeout).  usage: gpu_profile.sh <tag> [bench flags]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
tag=$1; shift
O=gpurun_out/$tag
mkdir -p $O
timeout 400 python bench.py "$@" > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc $?"
B="python bench.py --steps 60 --warmup 50 --no-cpu --no-episode --parity-envs 0 $*"
timeout 150 rocprofv3 --kernel-trace --stats -d $O/trace -o r -- python bench.py --steps 200 --warmup 50 --no-cpu --no-episode --parity-envs 0 "$@" > $O/trace_line.json 2> $O/trace.err; echo "trace rc $?"
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o r -- $B > $O/fetch_line.json 2> $O/fetch.err; echo "fetch rc $?"
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o r -- $B > $O/write_line.json 2> $O/write.err; echo "write rc $?"
timeout 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/sq -o r -- $B > $O/sq_line.json 2> $O/sq.err; echo "sq rc $?"
python tools/refresh_pmc.py $O profiles/${tag}_rocprofv3.md $O
find $O -name "*.db" -delete; find $O -type d -empty -delete; du -sh $O
