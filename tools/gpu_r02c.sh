#!/bin/bash
# round-2 checkpoint: citi_bike city-size tests + bench lines, then the CIM profile refresh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${1:-r02c}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_citi_bike.py -x -q > $O/pytest_cb.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest_cb.log
for v in "--envs 4096" "--envs 4096 --topology city.180s" "--envs 1024 --topology city.180s" "--envs 16384 --topology city.180s"; do
  f=$O/bench_cb_$(echo $v | tr -d ' -' | tr '.' '_')
  timeout 300 python bench.py --scenario citi_bike $v --no-cpu --steps 300 --warmup 50 > $f.json 2> $f.err
  echo "citi_bike $v: rc $? $(python -c "import json; d=json.load(open('$f.json')); print(round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4), 'ms spec', d['config']['specialized_kernels'], 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'tbar', round(d['config']['mean_ticks_per_env_step'],3))" 2>&1 | tail -1)"
done
timeout 200 rocprofv3 --kernel-trace --stats -d $O/cbtrace -o r -- python bench.py --scenario citi_bike --envs 4096 --topology city.180s --no-cpu --steps 200 --warmup 50 > $O/cbtrace_line.json 2> $O/cbtrace.err; echo "cb trace rc $?"
python - <<PY
import csv, glob
for f in glob.glob("$O/cbtrace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open("$O/cb_city_kernel_stats.md", "w") as fp:
        fp.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
        for r in rows[:12]:
            fp.write(f"| {r['Name'][:60]} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.1f} | {r['Percentage']} |\n")
    print(open("$O/cb_city_kernel_stats.md").read())
PY
find $O/cbtrace -name "*.db" -delete
