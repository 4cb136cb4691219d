#!/bin/bash
# round-end check on the GPU box: full -m gpu suite, smoke(), the default bench lines, the phase profile of the default launch form
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${1:-final}
mkdir -p $O
timeout 1700 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -1 $O/smoke.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
python -c "import json; d=json.load(open('$O/bench_default.json')); print(round(d['value']/1e6,1), 'M', round(d.get('value_end_to_end',0)/1e6,1), 'M e2e, frac', round(d['roofline']['frac'],3), 'parity', d['parity']['ok'], 'cpu', round(d['cpu_baseline']['value']/1e6,2))"
timeout 300 python bench.py --scenario citi_bike > $O/bench_cb_default.json 2> $O/bench_cb_default.err; echo "bench cb rc $?"
python -c "import json; d=json.load(open('$O/bench_cb_default.json')); print(round(d['value']/1e6,1), 'M', d['config']['workload'][:60], 'bounded', round(d['bounded_steps']['value']/1e6,1), 'cpu', d['cpu_baseline']['value'])"
timeout 200 python tools/phase_profile.py --specialized --step-mode 2 > $O/phase_m2.txt 2>&1; tail -18 $O/phase_m2.txt
