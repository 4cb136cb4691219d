#!/bin/bash
# round 4, call C: v2 loop test in both dtypes, forward tile rows 16 vs 32, kernel trace of the loop
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r04c; mkdir -p $O
timeout 600 python -m pytest tests/test_sampler.py tests/test_gpu_dqn.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
MRX_DQN_TILE=16 timeout 300 python -m pytest tests/test_gpu_dqn.py -m gpu -q -k "fused_dqn" > $O/pytest16.log 2>&1; echo "pytest16 rc $?"; tail -3 $O/pytest16.log
C="--policy dqn --collect --ring 8 --envs 8192 --no-cpu --repeats 3 --parity-envs 0 --steps 64 --warmup 16"
for t in 32 16; do for g in 1 2 3; do MRX_DQN_TILE=$t timeout 200 python bench.py $C --groups $g > $O/collect_t${t}_g$g.json 2> $O/collect_t${t}_g$g.err; echo "collect t$t g$g rc $?"; done; done
MRX_DQN_TILE=16 GPU_MAX_HW_QUEUES=8 timeout 200 python bench.py $C --groups 4 > $O/collect_t16_q8_g4.json 2> $O/collect_t16_q8_g4.err
MRX_DQN_TILE=16 timeout 200 python bench.py --policy dqn --ring 8 --envs 16384 --no-cpu --parity-envs 0 --secondary 0 --steps 200 --repeats 3 > $O/dqn_t16.json 2> $O/dqn_t16.err
MRX_DQN_TILE=32 timeout 200 python bench.py --policy dqn --ring 8 --envs 16384 --no-cpu --parity-envs 0 --secondary 0 --steps 200 --repeats 3 > $O/dqn_t32.json 2> $O/dqn_t32.err
for t in 32 16; do
MRX_DQN_TILE=$t timeout 300 rocprofv3 --kernel-trace --stats -d $O/ctrace$t -o r -- python bench.py $C --groups 2 --repeats 2 > $O/ctrace${t}_line.json 2> $O/ctrace$t.err; echo "ctrace$t rc $?"
python tools/rocprof_summary.py $(find $O/ctrace$t -name "r_results.db" | head -1) 2>&1 | head -14 > $O/collect_trace_t$t.md
done
find $O -name "*.db" -delete; find $O -type d -empty -delete
for f in $O/collect_t*.json $O/dqn_*.json; do python - "$f" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]/1e6,2), "M", round(d["value_min"]/1e6,1), round(d["value_max"]/1e6,1), "ms", round(d["ms_per_step"],4), "act_ms", d.get("roofline_policy",{}).get("kernel_ms"))
except Exception as e: print(sys.argv[1], "ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-600:])
P
done
