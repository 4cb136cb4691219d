#!/bin/bash
# timeline of the end-to-end leg with the progressive reset: where do the table blocks run, what happens to the step kernels
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04q; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
for v in "C 128 512" "F 128 0" "A 0 0"; do
  set -- $v
  (cd /tmp && GPU_MAX_HW_QUEUES=8 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$1 -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 60 --warmup 10 --repeats 1 --secondary 0 --no-cpu --parity-envs 0 --table-block $2 --table-waves $3 > $O/bench_$1.json 2> $O/bench_$1.err)
  f=$(find $O/trace_$1 -name '*kernel_trace.csv' | head -1)
  python - "$1" "$f" <<'PY' > $O/timeline_$1.txt 2>&1
import csv, sys
name, path = sys.argv[1], sys.argv[2]
rows = []
with open(path) as fp:
    for r in csv.DictReader(fp):
        rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0), r.get("Queue_Id", "")))
rows.sort(key=lambda x: x[1])
# the end-to-end leg starts at the LAST burst of reset kernels
resets = [i for i, r in enumerate(rows) if r[0].startswith("mrx_k_cim_reset")]
last = resets[-1]
i0 = last
while i0 - 1 in resets or (i0 > 0 and rows[i0 - 1][0].startswith("mrx_k_cim_reset")): i0 -= 1
first = [i for i in resets if rows[i][1] >= rows[last][1] - 20_000_000][0]
t0 = rows[first][1]
ep = [r for r in rows[first:]]
print(name, "kernels after the e2e reset:", len(ep))
print("-- reset / bounds / table dispatches: start ms, dur ms, grid, queue")
for k, s, e, g, q in ep:
    if "order_table" in k or "cim_reset" in k or "decision_bounds" in k:
        print("  %-28s %8.3f %8.3f %8d %s" % (k[:28], (s - t0) / 1e6, (e - s) / 1e6, g, q))
print("-- step kernels per 10 ms bin: count, mean dur us, max dur us")
bins = {}
for k, s, e, g, q in ep:
    if "cim_step" in k:
        b = int((s - t0) / 1e7)
        bins.setdefault(b, []).append((e - s) / 1e3)
for b in sorted(bins):
    v = bins[b]
    print("  %4d-%4d ms  n %5d  mean %7.1f  max %8.1f" % (b * 10, b * 10 + 10, len(v), sum(v) / len(v), max(v)))
end = max(e for k, s, e, g, q in ep if "cim_step" in k)
print("episode span ms: %.2f" % ((end - t0) / 1e6))
PY
  rm -rf $O/trace_$1
done
cat $O/timeline_C.txt | head -70; echo; head -60 $O/timeline_F.txt; echo; head -40 $O/timeline_A.txt
