#!/bin/bash
# SQ instruction counters of the citi_bike step kernel (rocprofv3 --pmc, kernel trace only).  usage: gpu_cb_sq.sh <tag> [bench flags]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
tag=$1; shift
O=gpurun_out/$tag
mkdir -p $O
B="python bench.py --scenario citi_bike --no-cpu --steps 100 --warmup 30 $*"
timeout 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/sq -o r -- $B > $O/sq_line.json 2> $O/sq.err; echo "sq rc $?"
timeout 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES -d $O/sq2 -o r -- $B > $O/sq2_line.json 2> $O/sq2.err; echo "sq2 rc $?"
python - <<PY
import sys
sys.path.insert(0, ".")
from tools.refresh_pmc import db, mean_counter
out = []
for name, ctrs in (("sq", "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"), ("sq2", "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES")):
    p = db("$O", name)
    if not p:
        continue
    for c in ctrs.split():
        v, n = mean_counter(p, "mrx_k_cb_step", c)
        out.append(f"{c:22s} {v:14.0f}  (mean over {n} dispatches)")
open("$O/cb_sq.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
find $O -name "*.db" -delete
