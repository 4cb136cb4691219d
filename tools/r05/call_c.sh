#!/bin/bash
O=gpurun_out/r05c; mkdir -p $O
export PYTHONPATH=$PWD
for v in new new2; do
  for n in 4096 8192; do
    MARO_AMD_LIB=$PWD/variants/${v}_prof/libmaro_amd.so MRX_DQN_TILE=16 timeout 300 python tools/dqn_phase_profile.py $n > $O/phases_${v}_$n.txt 2>&1
  done
done
for rep in 1 2; do
for v in new new2; do
  for g in 1 2; do
    MARO_AMD_LIB=$PWD/variants/$v/libmaro_amd.so timeout 600 python bench.py --policy dqn --collect --envs 8192 --ring 8 --steps 64 --warmup 16 --repeats 3 --groups $g --no-cpu --parity-envs 0 > $O/collect_${v}_g${g}_r$rep.json 2> $O/collect_${v}_g${g}_r$rep.err
  done
done
done
for v in new new2; do
MARO_AMD_LIB=$PWD/variants/$v/libmaro_amd.so timeout 600 python bench.py --policy dqn --envs 16384 --ring 8 --steps 200 --warmup 20 --repeats 3 --no-cpu --parity-envs 0 --no-episode > $O/dqn_$v.json 2> $O/dqn_$v.err
done
MARO_AMD_LIB=$PWD/variants/new2/libmaro_amd.so timeout 600 python -m pytest tests/test_gpu_dqn.py tests/test_sampler.py tests/test_policy.py -m gpu -x -q > $O/pytest_dqn.log 2>&1; echo "pytest rc $?" >> $O/pytest_dqn.log
python - <<P
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d["value"]/1e6,2), "M", round(d["ms_per_step"]*1e3,1), "us")
    except Exception as e: print(f, "FAILED", e)
P
tail -3 $O/pytest_dqn.log; cat $O/phases_*.txt | grep -v amdgpu.ids
