# city.800s A/B: replay kernel beside / after the in-tick kernel (mrx_cb_set_replay_overlap) x step budget.  BUDGETS / TESTS env vars.
export TMPDIR=/tmp
if [ "${TESTS:-1}" = "1" ]; then timeout 900 python -m pytest tests/test_gpu_citi_bike.py tests/test_gpu_specialized.py -x -q 2>&1 | tail -5; fi
C="--scenario citi_bike --no-cpu --topology city.800s --envs 4096 --durations 2880 --steps 900 --warmup 300 --bounded-budget 0 --specialize 1"
for ov in ${OVERLAPS:-0 1}; do for b in ${BUDGETS:-12 16 24 32 48}; do
  timeout 200 python bench.py $C --step-budget $b --replay-overlap $ov > gpurun_out/c800_ov${ov}_b$b.json 2>gpurun_out/c800_ov${ov}_b$b.err
  echo "overlap $ov budget $b: $(python tools/show_line.py gpurun_out/c800_ov${ov}_b$b.json 2>&1 | head -1)"
done; done
