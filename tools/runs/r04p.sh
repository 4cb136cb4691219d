#!/bin/bash
# experiment: order table in blocks behind the first steps (mrx_cim_set_progressive_reset) — end-to-end rate
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04p; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 600 python -m pytest tests/test_gpu_progressive_reset.py -x -q -m gpu > $O/pytest_new.log 2>&1; echo "new tests rc=$?" >> $O/summary.txt
tail -3 $O/pytest_new.log >> $O/summary.txt
run() { # name, env, args...
  name=$1; shift; envs=$1; shift
  env $envs timeout 300 python bench.py --gpus 1 --steps 150 --warmup 30 --repeats 2 --secondary 0 --no-cpu --parity-envs 0 "$@" > $O/$name.json 2> $O/$name.err
  python - "$name" $O/$name.json <<'PY' >> $O/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], "steady %.1f M"%(d["value"]/1e6), "e2e %.1f M"%(d.get("value_end_to_end",0)/1e6), "ep_s %.4f"%d["end_to_end"]["seconds"], "batch_steps", d["end_to_end"]["batch_steps"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run A_plain "X=1" --table-block 0
run B_b128_w512 "X=1" --table-block 128 --table-waves 512
run C_q8_b128_w512 "GPU_MAX_HW_QUEUES=8" --table-block 128 --table-waves 512
run D_q8_b128_w256 "GPU_MAX_HW_QUEUES=8" --table-block 128 --table-waves 256
run E_q8_b128_w1024 "GPU_MAX_HW_QUEUES=8" --table-block 128 --table-waves 1024
run F_q8_b128_full "GPU_MAX_HW_QUEUES=8" --table-block 128 --table-waves 0
run G_q8_b64_w512 "GPU_MAX_HW_QUEUES=8" --table-block 64 --table-waves 512
run H_q8_b280_w512 "GPU_MAX_HW_QUEUES=8" --table-block 280 --table-waves 512
run I_q8_plain "GPU_MAX_HW_QUEUES=8" --table-block 0
run J_b128_w128 "X=1" --table-block 128 --table-waves 128
# parity with the progressive reset on (the parity leg resets the same engines)
timeout 400 python bench.py --gpus 1 --steps 100 --warmup 20 --repeats 1 --secondary 0 --no-cpu --parity-envs 32 --table-block 128 --table-waves 512 > $O/parity.json 2> $O/parity.err
python -c "
import json; d=json.loads(open('$O/parity.json').read().strip().splitlines()[-1]); print('parity', d.get('parity',{}).get('ok'), 'e2e %.1f M'%(d['value_end_to_end']/1e6))" >> $O/summary.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_step_modes.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/pytest_old.log 2>&1; echo "old tests rc=$?" >> $O/summary.txt
tail -2 $O/pytest_old.log >> $O/summary.txt
cat $O/summary.txt
