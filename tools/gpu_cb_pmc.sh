#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the citi_bike step kernel (separate --pmc passes, kernel trace only).  usage: gpu_cb_pmc.sh <tag> [bench flags]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
tag=$1; shift
O=gpurun_out/$tag
mkdir -p $O
B="python bench.py --scenario citi_bike --no-cpu --steps 100 --warmup 30 --bounded-budget 0 $*"
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o r -- $B > $O/fetch_line.json 2> $O/fetch.err; echo "fetch rc $?"
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o r -- $B > $O/write_line.json 2> $O/write.err; echo "write rc $?"
python - <<PY
import sys
sys.path.insert(0, ".")
from tools.refresh_pmc import db, mean_counter
f, nf = mean_counter(db("$O", "fetch"), "mrx_k_cb_step", "FETCH_SIZE")
w, nw = mean_counter(db("$O", "write"), "mrx_k_cb_step", "WRITE_SIZE")
txt = f"mrx_k_cb_step [$*]: FETCH_SIZE {f:.1f} KiB (x2 gfx950 correction = {2*f*1.024/1000:.2f} MB) + WRITE_SIZE {w:.1f} KiB ({w*1.024/1000:.2f} MB) per launch ({nf}/{nw} dispatches)"
open("$O/cb_pmc.txt", "w").write(txt + "\n")
print(txt)
PY
find $O -name "*.db" -delete
