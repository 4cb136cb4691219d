#!/bin/bash
# round 2, first GPU call: correctness of the new launch forms + a sweep of step mode x groups (run through gpurun)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r02a
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_step_modes.py -x -q > $O/pytest.log 2>&1
echo "pytest rc $?" >> $O/pytest.log
tail -3 $O/pytest.log
for mode in 3 2 1; do
  for g in 1 2 3; do
    timeout 200 python bench.py --no-cpu --steps 300 --warmup 100 --step-mode $mode --groups $g > $O/bench_m${mode}_g${g}.json 2> $O/bench_m${mode}_g${g}.err
    echo "m$mode g$g rc $? $(python -c "import json,sys; d=json.load(open('$O/bench_m${mode}_g${g}.json')); print(round(d['value']/1e6,1), 'M', round(d['ms_per_step'],4), 'ms', d['config']['step_mode'], round(d['roofline']['kernel_ms'],4))" 2>&1 | tail -1)"
  done
done
for w in 4 6; do
  MRX_CIM_PIPE_WAVES_PER_CU=$w timeout 200 python bench.py --no-cpu --steps 300 --warmup 100 --step-mode 3 --groups 1 > $O/bench_m3_g1_w$w.json 2> $O/bench_m3_g1_w$w.err
  echo "m3 g1 waves/CU $w: $(python -c "import json; d=json.load(open('$O/bench_m3_g1_w$w.json')); print(round(d['value']/1e6,1), 'M')" 2>&1 | tail -1)"
done
timeout 300 python bench.py --no-cpu --steps 300 --warmup 100 --step-mode 3 --groups 1 --envs 65536 > $O/bench_m3_g1_64k.json 2> $O/bench_m3_g1_64k.err
echo "m3 g1 65536 envs: $(python -c "import json; d=json.load(open('$O/bench_m3_g1_64k.json')); print(round(d['value']/1e6,1), 'M')" 2>&1 | tail -1)"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o r -- python bench.py --steps 100 --warmup 50 --no-cpu --step-mode 3 --groups 1 > $O/trace_bench.json 2> $O/trace.err
python tools/rocprof_summary.py $O/trace/*/r_results.db > $O/trace_summary.md 2>&1 || python tools/rocprof_summary.py $O/trace/r_results.db > $O/trace_summary.md 2>&1
head -30 $O/trace_summary.md
