#!/bin/bash
# round 4, call E: citi_bike city.800s after the LDS diet, DQN forward phases in situ (tile 16 / 32, 4096 / 8192 envs)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r04e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_citi_bike.py tests/test_gpu_citi_bike_api.py tests/test_gpu_specialized.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
B="--scenario citi_bike --no-cpu --bounded-budget 0 --repeats 3"
timeout 300 python bench.py $B --topology city.800s --envs 4096 --durations 2880 --steps 900 --warmup 300 --step-budget 64 --parity-envs 0 > $O/cb_city.json 2> $O/cb_city.err; echo "cb city rc $?"
timeout 300 python bench.py $B --topology city.800s --envs 4096 --durations 2880 --steps 900 --warmup 300 --step-budget 0 --parity-envs 0 > $O/cb_city_nobudget.json 2> $O/cb_city_nobudget.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/cbtrace -o r -- python bench.py $B --topology city.800s --envs 4096 --durations 2880 --steps 900 --warmup 300 --step-budget 64 --parity-envs 0 --repeats 1 > $O/cbtrace_line.json 2> $O/cbtrace.err
python tools/rocprof_summary.py $(find $O/cbtrace -name "r_results.db" | head -1) 2>&1 | head -12 | cut -c1-200 > $O/cb_city_trace.md
(cd maro_amd/csrc && hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value -DMRX_DQN_PROFILE -o /tmp/libmaro_amd_prof.so cim_engine.hip cb_engine.hip 2> /tmp/prof_build.err); echo "prof build rc $?"
for t in 16 32; do for n in 4096 8192; do MRX_DQN_TILE=$t MARO_AMD_LIB=/tmp/libmaro_amd_prof.so timeout 200 python tools/dqn_phase_profile.py $n >> $O/dqn_phases.txt 2>> $O/dqn_phases.err; done; done
cat $O/dqn_phases.txt
find $O -name "*.db" -delete; find $O -type d -empty -delete
for f in $O/cb_city*.json; do python - "$f" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]/1e6,2), "M", round(d["value_min"]/1e6,1), round(d["value_max"]/1e6,1), "ms", round(d["ms_per_step"],4))
except Exception as e: print(sys.argv[1], "ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-600:])
P
done
cat $O/cb_city_trace.md
