#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r02m
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_bench_parity.py -x -q > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
timeout 300 python bench.py --cpu-seconds 4 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -3 $O/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r02m/bench.json"))
print(round(d["value"]/1e6,1), "M steady;", round(d.get("value_end_to_end",0)/1e6,1), "M end-to-end;", d["ms_per_step"], d["config"]["mean_tick_at_window_start"], d["config"]["mean_ticks_per_env_step"])
print(d.get("end_to_end")); print(d.get("parity")); print(d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["algorithmic_frac"], d["roofline"]["kernel_ms"], d["roofline"]["launches_in_flight"])
print(d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline_reference"))
PY
