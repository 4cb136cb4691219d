# city.800s: kernel timeline of a few batch steps (start offsets, durations) — are the two wave kernels side by side?
export TMPDIR=/tmp; cd /tmp
C="--scenario citi_bike --no-cpu --topology city.800s --envs 4096 --durations 2880 --steps 300 --warmup 300 --bounded-budget 0 --specialize 1 --step-budget ${BUDGET:-24}"
timeout 300 rocprofv3 --kernel-trace -d /tmp/tl -o r -- python /root/repo/bench.py $C --replay-overlap ${OV:-1} > /dev/null 2>/tmp/tl.err
f=$(find /tmp/tl -name "r_results.db" | head -1)
python - "$f" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
names = dict(c.execute("select id, kernel_name from rocpd_info_kernel_symbol"))
rows = [(names.get(k, str(k)), s, e) for k, s, e in c.execute("select kernel_id, start, end from rocpd_kernel_dispatch")]
rows = sorted((r for r in rows if r[0].startswith("mrx_k_cb")), key=lambda r: r[1])
mid = len(rows) * 3 // 4
t0 = rows[mid][1]
for n, s, e in rows[mid:mid + 40]:
    print(f"{n[:28]:28s} start {(s-t0)/1000:9.1f} us  end {(e-t0)/1000:9.1f} us  dur {(e-s)/1000:7.1f}")
# per batch step: first kernel start -> next step's first kernel start
pol = [r for r in rows if "random_policy" in r[0]]
gaps = [(b[1] - a[1]) / 1000 for a, b in zip(pol[300:], pol[301:])]
gaps.sort()
print("step period us: median %.1f  p10 %.1f  p90 %.1f" % (gaps[len(gaps)//2], gaps[len(gaps)//10], gaps[len(gaps)*9//10]))
for k in ("classify", "step_wave", "replay_wave", "random_policy"):
    d = sorted((e - s) / 1000 for n, s, e in rows if k in n)
    if d: print(f"{k}: n {len(d)} mean {sum(d)/len(d):.1f} median {d[len(d)//2]:.1f} p90 {d[len(d)*9//10]:.1f} max {d[-1]:.1f}")
PY
