export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_citi_bike.py tests/test_gpu_specialized.py -x -q 2>&1 | tail -5
C="--scenario citi_bike --no-cpu --topology city.800s --envs 4096 --durations 2880 --steps 900 --warmup 300 --bounded-budget 0 --specialize 1"
for ov in 0 1; do for b in 12 16 24 32 48; do
  timeout 200 python bench.py $C --step-budget $b --replay-overlap $ov > gpurun_out/c800_ov${ov}_b$b.json 2>gpurun_out/c800_ov${ov}_b$b.err
  echo "overlap $ov budget $b: $(python tools/show_line.py gpurun_out/c800_ov${ov}_b$b.json 2>&1 | head -1)"
done; done
