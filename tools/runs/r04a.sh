#!/bin/bash
# round 4, call A: the GPU suite, the driver's bench command, and scheduling experiments (collect groups, step-kernel queues)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r04a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "not default_line" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc $?"
for g in 1 2 3; do timeout 200 python bench.py --policy dqn --collect --ring 8 --envs 8192 --groups $g --no-cpu --parity-envs 0 --steps 64 --warmup 16 --repeats 3 > $O/collect_g$g.json 2> $O/collect_g$g.err; echo "collect g$g rc $?"; done
S="--no-cpu --no-episode --parity-envs 0 --secondary 0 --steps 300 --repeats 3"
timeout 200 python bench.py $S > $O/step_g3.json 2> $O/step_g3.err
GPU_MAX_HW_QUEUES=8 timeout 200 python bench.py $S --groups 4 > $O/step_q8_g4.json 2> $O/step_q8_g4.err
GPU_MAX_HW_QUEUES=8 timeout 200 python bench.py $S --groups 6 > $O/step_q8_g6.json 2> $O/step_q8_g6.err
GPU_MAX_HW_QUEUES=8 timeout 200 python bench.py $S --groups 3 > $O/step_q8_g3.json 2> $O/step_q8_g3.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/ctrace -o r -- python bench.py --policy dqn --collect --ring 8 --envs 8192 --no-cpu --parity-envs 0 --steps 64 --warmup 16 --repeats 2 > $O/ctrace_line.json 2> $O/ctrace.err; echo "ctrace rc $?"
python tools/rocprof_summary.py $(find $O/ctrace -name "r_results.db" | head -1) > $O/collect_trace.md 2>&1
find $O -name "*.db" -delete; find $O -type d -empty -delete
for f in $O/collect_g*.json $O/step_*.json; do python - "$f" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]/1e6,1), "M", d.get("values") and [round(v/1e6,1) for v in d["values"]], d["ms_per_step"])
except Exception as e: print(sys.argv[1], "ERR", e)
P
done
