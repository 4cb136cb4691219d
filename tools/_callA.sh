cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/r03x; mkdir -p $O
B="python bench.py --steps 300 --warmup 20 --no-cpu --no-episode --parity-envs 0 --repeats 3"
run(){ name=$1; shift; timeout 150 "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc $?"; }
run base $B
MRX_CIM_LPT=0 run lpt0 $B
run m5 $B --step-mode 5
MRX_CIM_LPT=0 run m5lpt0 $B --step-mode 5
run g4 $B --groups 4
run g6 $B --groups 6
GPU_MAX_HW_QUEUES=8 run g4q8 $B --groups 4
GPU_MAX_HW_QUEUES=8 run g6q8 $B --groups 6
run base2 $B
C="python bench.py --policy dqn --collect --ring 8 --no-cpu --steps 128 --warmup 32 --repeats 3"
run col16 $C
run col8 $C --envs 8192
timeout 200 rocprofv3 --kernel-trace --stats -d $O/ctr -o r -- python bench.py --policy dqn --collect --ring 8 --no-cpu --groups 1 --steps 128 --warmup 32 --repeats 1 > $O/ctr.json 2> $O/ctr.err; echo "ctr rc $?"
python tools/rocprof_summary.py $O/ctr/r_results.db > $O/ctr.md 2>&1
timeout 500 python -m pytest tests/test_gpu_step_modes.py tests/test_sampler.py tests/test_gpu_vector_env.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
find $O -name "*.db" -delete; find $O -type d -empty -delete
for f in base lpt0 m5 m5lpt0 g4 g6 g4q8 g6q8 base2 col16 col8; do python -c "
import json,sys
try:
    d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['value']/1e6,2), d.get('values'))
except Exception as e: print('$f','ERR',e)
"; done
