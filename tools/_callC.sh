cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/r03z4; mkdir -p $O
run(){ name=$1; shift; timeout 200 "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc $?"; }
C="python bench.py --policy dqn --collect --ring 8 --no-cpu --steps 128 --warmup 32 --repeats 3"
run col16g2 $C --groups 2
GPU_MAX_HW_QUEUES=8 run col16g6 $C --groups 6
GPU_MAX_HW_QUEUES=8 run col16g4 $C --groups 4
run col32k $C --envs 32768
run dqn python bench.py --policy dqn --no-cpu
run dqngr python bench.py --policy dqn --no-cpu --graphs 1
for f in col16g2 col16g6 col16g4 col32k dqn dqngr; do python -c "
import json,sys
try:
    d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['value']/1e6,2), d['ms_per_step'], d.get('experiences_per_s'), d['config'].get('host_enqueue_ms_per_step'))
except Exception as e: print('$f','ERR',e)
"; done
