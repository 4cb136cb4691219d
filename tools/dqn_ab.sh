# usage: bash tools/dqn_ab.sh <outdir> [tests]   — the DQN act launch on one box: (GPU tests,) phase profile (profiling build in
# variants/prof/, if present), the collection loop of config 5 (2 groups and 1 group) and act -> step at 16 384 envs
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/$1; mkdir -p $O
if [ "$2" = tests ]; then timeout 900 python -m pytest tests/test_gpu_dqn.py tests/test_sampler.py -x -q -m gpu > $O/tests.txt 2>&1; echo "tests rc $?"; tail -3 $O/tests.txt; fi
if [ -f variants/prof/libmaro_amd.so ]; then
  MARO_AMD_LIB=variants/prof/libmaro_amd.so timeout 300 python tools/dqn_phase_profile.py 4096 > $O/dqn_phase.txt 2>&1; echo "phase rc $?"; grep -v amdgpu.ids $O/dqn_phase.txt | head -12
fi
line() { python - "$1" <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    rp = d.get("roofline_policy") or {}
    print(sys.argv[1], round(d["value"] / 1e6, 2), "M  ms/step", round(d["ms_per_step"], 4), " policy frac", round(rp.get("frac") or 0, 3), "act us", round((rp.get("kernel_ms") or 0) * 1e3, 1), "parity", (d.get("parity") or {}).get("ok"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
P
}
CF="--policy dqn --collect --ring 8 --envs 8192 --no-cpu --steps 64 --warmup 16"
timeout 600 python bench.py $CF --groups 2 --parity-envs ${PARITY:-0} > $O/collect_g2.json 2> $O/collect_g2.err; line $O/collect_g2.json
timeout 600 python bench.py $CF --groups 1 --parity-envs 0 > $O/collect_g1.json 2> $O/collect_g1.err; line $O/collect_g1.json
for e in $EXTRA; do timeout 600 python bench.py $CF --groups $e --parity-envs 0 > $O/collect_g$e.json 2> $O/collect_g$e.err; line $O/collect_g$e.json; done
