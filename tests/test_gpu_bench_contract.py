"""bench.py's output contract (one JSON line on stdout with the driver's keys plus `roofline` and `cpu_baseline`), checked on a
tiny batch so it costs seconds."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*extra):
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "3", *extra],
                         capture_output=True, text=True, cwd=REPO, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("extra", [("--envs", "192", "--cpu-seconds", "2"), ("--scenario", "citi_bike", "--envs", "128", "--no-cpu"),
                                   ("--envs", "192", "--policy", "dqn", "--no-cpu")])
def test_bench_line_contract(extra):
    j = run_bench(*extra)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 6 and j["warmup"] == 3 and j["higher_is_better"] is True and j["scaling"] == "weak"
    assert j["vs_baseline"] is None and j["data"] == "synthetic" and j["value"] > 0 and j["ms_per_step"] > 0
    assert "workload" in j["config"] and not any(k in j["config"] for k in ("model", "seq_len", "global_batch"))
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert "traffic" in r and r["kernel"].startswith("mrx_k_")
    if "--no-cpu" not in extra:
        c = j["cpu_baseline"]
        assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == j["unit"] and "sample" in c
    if "dqn" in extra:
        assert j["roofline_policy"]["bound"] == "mfma" and j["roofline_policy"]["unit"] == "TFLOP/s"
