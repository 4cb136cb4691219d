#!/usr/bin/env python3
"""Print the fields of a bench.py JSON line that matter at a glance (the headline, every secondary leg: value, roofline fraction,
parity, cpu baseline kind / value).    python tools/show_line.py gpurun_out/<tag>/bench_line.json"""
import json
import sys


def one(name, d):
    if not isinstance(d, dict) or "value" not in d:
        print(f"{name}: {str(d)[:200]}")
        return
    rf, par, cb = d.get("roofline") or {}, d.get("parity") or {}, d.get("cpu_baseline") or {}
    print(f"{name}: {d['value'] / 1e6:.2f} M {d.get('unit', '')}  ms/step {d.get('ms_per_step', 0):.4f}  frac {rf.get('frac')}  parity {par.get('ok')}"
          f"  cpu[{cb.get('kind')}] {cb.get('value')}")
    for k in ("value_end_to_end", "value_sustained", "roofline_policy", "object_api"):
        if k in d:
            v = d[k]
            print(f"    {k}: {json.dumps(v)[:260] if isinstance(v, dict) else v}")
    if "cpu_baseline_reference" in d:
        print(f"    cpu_baseline_reference: {json.dumps(d['cpu_baseline_reference'])[:200]}")


def main():
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    one("headline", d) if "metric" in d else one("line", d)
    for k, v in (d.get("secondary") or {}).items():
        one("secondary." + k, v)
    print("gpu_seconds_total", d.get("gpu_seconds_total"))


if __name__ == "__main__":
    main()
