cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/r03z3; mkdir -p $O
run(){ name=$1; shift; timeout 200 "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc $?"; }
C="python bench.py --policy dqn --collect --ring 8 --no-cpu --steps 128 --warmup 32 --repeats 3"
run col16 $C
run col8 $C --envs 8192
run col16g1 $C --groups 1
run col16s256 python bench.py --policy dqn --collect --ring 8 --no-cpu --steps 256 --warmup 32 --repeats 3
timeout 600 python -m pytest tests/test_sampler.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for f in col16 col8 col16g1 col16s256; do python -c "
import json,sys
try:
    d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['value']/1e6,2), d['ms_per_step'], d.get('experiences_per_s'))
except Exception as e: print('$f','ERR',e)
"; done
