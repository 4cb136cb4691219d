cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/r03y; mkdir -p $O
B="python bench.py --no-cpu"
run(){ name=$1; shift; timeout 200 "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc $?"; }
run g3 $B --groups 3
run g4 $B --groups 4
run g5 $B --groups 5
run g4b $B --groups 4
run dqn4 python bench.py --policy dqn --no-cpu --groups 4
run col4 python bench.py --policy dqn --collect --ring 8 --no-cpu --steps 128 --warmup 32 --repeats 3 --groups 4
timeout 600 python -m pytest tests/test_gpu_step_modes.py tests/test_sampler.py tests/test_gpu_vector_env.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for f in g3 g4 g5 g4b dqn4 col4; do python -c "
import json,sys
try:
    d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['value']/1e6,2), d.get('values'), d.get('value_end_to_end'), (d.get('parity') or {}).get('ok'), d['config'].get('reset_ms_whole_batch'))
except Exception as e: print('$f','ERR',e)
"; done
