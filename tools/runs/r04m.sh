#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r04m; mkdir -p $O
B="--scenario citi_bike --no-cpu --bounded-budget 0 --repeats 3 --steps 400 --warmup 100 --parity-envs 0"
for b in 0 12 16 24 32 48; do timeout 200 python bench.py $B --step-budget $b > $O/cb_b$b.json 2> $O/cb_b$b.err; done
for f in $O/cb_b*.json; do python - "$f" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]/1e6,2), "M", round(d["value_min"]/1e6,1), round(d["value_max"]/1e6,1), "ms", round(d["ms_per_step"],4))
except Exception as e: print(sys.argv[1], "ERR", e)
P
done
