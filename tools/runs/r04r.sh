#!/bin/bash
# experiment 2: step-chain kernels at s_setprio 3, table blocks in the background
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04r; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 600 python -m pytest tests/test_gpu_progressive_reset.py -x -q -m gpu > $O/pytest_new.log 2>&1; echo "new tests rc=$?" >> $O/summary.txt
run() { name=$1; shift; envs=$1; shift
  env $envs timeout 300 python bench.py --gpus 1 --steps 150 --warmup 30 --repeats 2 --secondary 0 --no-cpu --parity-envs 0 "$@" > $O/$name.json 2> $O/$name.err
  python - "$name" $O/$name.json <<'PY' >> $O/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], "steady %.1f M"%(d["value"]/1e6), "e2e %.1f M"%(d.get("value_end_to_end",0)/1e6), "ep_s %.4f"%d["end_to_end"]["seconds"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run A_plain "GPU_MAX_HW_QUEUES=8" --table-block 0
run A2_plain_q4 "X=1" --table-block 0
run F_full "GPU_MAX_HW_QUEUES=8" --table-block 128 --table-waves 0
run W512 "GPU_MAX_HW_QUEUES=8" --table-block 128 --table-waves 512
run W1024 "GPU_MAX_HW_QUEUES=8" --table-block 128 --table-waves 1024
run W1365 "GPU_MAX_HW_QUEUES=8" --table-block 128 --table-waves 1365
run W2048 "GPU_MAX_HW_QUEUES=8" --table-block 128 --table-waves 2048
run W1024_b64 "GPU_MAX_HW_QUEUES=8" --table-block 64 --table-waves 1024
run W1024_b280 "GPU_MAX_HW_QUEUES=8" --table-block 280 --table-waves 1024
run W1024_q4 "X=1" --table-block 128 --table-waves 1024
run A3_plain "GPU_MAX_HW_QUEUES=8" --table-block 0
cat $O/summary.txt
