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
                                   ("--envs", "192", "--policy", "dqn", "--no-cpu"), ("--envs", "192", "--agent", "fused", "--no-cpu")])
def test_bench_line_contract(extra):
    j = run_bench(*extra)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 6 and j["warmup"] == 3 and j["higher_is_better"] is True and j["scaling"] == "weak"
    assert j["vs_baseline"] is None and j["data"] == "synthetic" and j["value"] > 0 and j["ms_per_step"] > 0
    assert "workload" in j["config"] and not any(k in j["config"] for k in ("model", "seq_len", "global_batch"))
    r = j["roofline"]
    # frac is measured bytes / time or null (no PMC record of this build and batch size) — never a formula, never above 1
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and "basis" in r
    assert (r["frac"] is None and r["achieved"] is None) or (abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1)
    assert "traffic" in r and r["kernel"].startswith("mrx_k_") and "algorithmic_frac" not in r
    if "--no-cpu" not in extra:
        c = j["cpu_baseline"]
        assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == j["unit"] and "sample" in c
    if "dqn" in extra:
        assert j["roofline_policy"]["bound"] == "mfma" and j["roofline_policy"]["unit"] == "TFLOP/s"
    if "fused" in extra:   # the agent answered inside the step kernel: one launch per step, the unsorted form, the same parity replay
        assert "inside the step kernel" in j["config"]["agent"] and j["config"]["step_mode"] == 1
    if extra[:2] == ("--envs", "192") and "dqn" not in extra:   # the CIM headline line: end-to-end leg + oracle parity replay
        assert j["value_end_to_end"] > 0 and j["end_to_end"]["env_steps"] > 0 and j["end_to_end"]["reset_ms_synchronised"] > 0
        assert j["parity"]["ok"] is True and j["parity"]["envs_checked"] >= 60 and j["parity"]["env_steps_checked"] > 1000
        assert j["config"]["mean_tick_at_window_start"] >= 300 and "survey_formula" in r and "secondary" not in j   # (--envs given: headline only)
    if "citi_bike" in extra:
        assert j["parity"]["ok"] is True and j["parity"]["env_steps_checked"] > 100, j["parity"]


def test_bench_default_line_carries_configs_4_and_5():
    """The driver's exact command: after the headline, BASELINE configs 4 (citi_bike toy.3s_4t, 4096 envs) and 5 (the DQN collection
    loop, 8192 envs) run as short legs and ride in the same JSON line, each with parity, cpu_baseline and an honest roofline."""
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--cpu-seconds", "4"],
                         capture_output=True, text=True, cwd=REPO, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["parity"]["ok"] is True and j["config"]["envs_per_gpu"] == 16384
    sec = j["secondary"]
    c4, c5 = sec["citi_bike_config4"], sec["collect_config5"]
    assert "toy.3s_4t" in c4["metric"] and c4["config"]["envs_per_gpu"] == 4096 and c4["value"] > 1e6 and c4["steps"] == 20
    assert c5["config"]["envs_per_gpu"] == 8192 and c5["value"] > 1e6 and c5["experiences_per_s"] > 0
    for leg in (c4, c5):
        assert leg["parity"]["ok"] is True, leg["parity"]
        assert leg["cpu_baseline"]["value"] > 0 and leg["ms_per_step"] > 0
        fr = leg["roofline"]["frac"]
        assert fr is None or 0 < fr < 1
    assert c5["parity"]["elements_checked"] > 500 and c5["parity"]["rewards_checked"] > 0 and c5["parity"]["policy_choices_checked"] > 100
    assert c5["roofline_policy"]["bound"] == "mfma" and 0 < c5["roofline_policy"]["frac"] < 1
    # round 6: end to end and sustained on the headline (also under config), config 4 whole on one GPU, the object-API leg
    assert j["value_end_to_end"] > 0 and j["value_sustained"] > 0 and j["config"]["value_sustained"] == j["value_sustained"]
    assert j["sustained"]["resets"] >= 6 and j["sustained"]["seconds"] > 0.5
    w4 = sec["citi_bike_config4_one_gpu"]
    assert w4["config"]["envs_per_gpu"] == 32768 and w4["parity"]["ok"] is True and w4["value"] > c4["value"]
    c8 = sec["citi_bike_city800"]   # the reference's own topology size, sustained (its own steps / warmup: a window over several decision ticks)
    assert "city.800s" in c8["metric"] and c8["config"]["envs_per_gpu"] == 4096 and c8["steps"] == 900 and c8["parity"]["ok"] is True and c8["value"] > 1e7, c8
    obj = sec["object_api"]
    assert obj["parity"]["ok"] is True and obj["value"] > 5e4 and obj["value_default_gc"]["value"] > 0 and obj["env_view"]["value"] > 100, obj
    ref = j.get("cpu_baseline_reference")
    if os.path.exists(os.path.join(REPO, "oracle", "_ref", "maro_ref.tgz")):     # the shipped reference build: timed live on this box
        assert ref["kind"] == "reference" and ref["measured"].startswith("live") and ref["vector_env"]["value"] > 0, ref
        assert c4["cpu_baseline"]["kind"] == "reference" and c4["cpu_baseline"]["measured"].startswith("live"), c4["cpu_baseline"]   # the reference's citi_bike Env
        assert obj["cpu_baseline"]["kind"] == "reference"


@pytest.mark.parametrize("scenario,world", [("cim", 2), ("citi_bike", 4), ("cim", 8), ("citi_bike", 8)])
def test_bench_multi_rank_path_under_gloo(scenario, world):
    """`bench.py --gpus N` as the driver launches it (torch.distributed.run, one process per rank), with the test hooks
    MRX_BENCH_BACKEND=gloo + MRX_BENCH_DEVICE=0 so that a 1-GPU box can run it (RCCL refuses several ranks on one device):
    barriers, max-over-ranks timing, summed counts, the grouped send/recv trajectory gather and rank != 0 waiting for build()."""
    env = dict(os.environ, MRX_BENCH_BACKEND="gloo", MRX_BENCH_DEVICE="0", MASTER_ADDR="127.0.0.1")
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", str(world), "--steps", "8", "--warmup", "3",
           "--scenario", scenario, "--envs", "256" if scenario == "cim" else "128", "--no-cpu"]
    if scenario == "cim":
        cmd += ["--preroll-ticks", "20", "--durations", "260", "--repeats", "2", "--gather-every", "16", "--no-episode", "--parity-envs", "0"]
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=REPO, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == world and j["value"] > 0 and j["scaling"] == "weak"
    assert j["config"]["trajectory_gather_ms_32_steps"] > 0
    if scenario == "cim":
        # whole-job aggregate: every rank's env-steps summed over the slowest rank's time
        r = j["ranks"]
        assert len(r["env_steps"]) == world and all(x > 0 for x in r["env_steps"])
        assert abs(j["value"] - sum(r["env_steps"]) / max(r["seconds"])) <= 1e-6 * j["value"]
        # per env and step (decision int32[8], action int32[4], metrics int64[3], done u8, observation f32[22 x 7]) ~ 700 B
        n, T = 256, 32
        assert j["config"]["trajectory_gather_bytes_per_rank"] == T * n * (8 * 4 + 4 * 4 + 3 * 8 + 1 + 22 * 7 * 4)
        ge = j["config"]["trajectory_gather_every"]
        assert ge["steps_per_rollout"] == 16 and ge["gather_ms"] > 0 and 0 < ge["gather_share"] < 1
        assert ge["bytes_per_rank"] == 16 * n * (8 * 4 + 4 * 4 + 3 * 8 + 1 + 22 * 7 * 4)


def test_bench_default_command_on_two_ranks_under_gloo():
    """The driver's N > 1 command as it is issued (`bench.py --gpus N --steps K --warmup W`, nothing else): the headline AND both
    secondary legs run on every rank (barriers, max-over-ranks times, summed counts, the trajectory / experience gathers and the
    policy broadcast of config 5), and rank 0 prints one line whose secondary values are whole-job aggregates."""
    env = dict(os.environ, MRX_BENCH_BACKEND="gloo", MRX_BENCH_DEVICE="0", MASTER_ADDR="127.0.0.1")
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "4"]
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=REPO, timeout=1500, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["envs_per_gpu"] == 16384 and len(j["ranks"]["env_steps"]) == 2
    c4, c5 = j["secondary"]["citi_bike_config4"], j["secondary"]["collect_config5"]
    assert c4["n_gpus"] == 2 and c4["config"]["envs_per_gpu"] == 4096 and c4["value"] > 1e6 and c4["config"]["trajectory_gather_ms_32_steps"] > 0
    assert c5["n_gpus"] == 2 and c5["config"]["envs_per_gpu"] == 8192 and c5["value"] > 1e6
    assert c5["config"]["experience_gather"]["experiences"] > 0 and c5["config"]["policy_broadcast"]["bytes"] > 1e6
    assert c4["parity"]["ok"] is True and c5["parity"]["ok"] is True


def test_bench_collect_multi_rank_path_under_gloo():
    """`bench.py --policy dqn --collect --gpus 2`: every rank runs the batched EnvSampler loop over its env shard, the value is the
    whole job's, and the learner-side collection (gather_experiences_to_learner) joins the ranks' experiences on rank 0."""
    env = dict(os.environ, MRX_BENCH_BACKEND="gloo", MRX_BENCH_DEVICE="0", MASTER_ADDR="127.0.0.1")
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "2", "--policy", "dqn", "--collect", "--ring", "8", "--steps", "24",
           "--warmup", "4", "--repeats", "1", "--envs", "192", "--durations", "260", "--preroll-ticks", "40", "--no-cpu"]
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=REPO, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout
    j = json.loads(lines[0])
    g = j["config"]["experience_gather"]
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["experiences_per_s"] > 0
    assert g["experiences"] > 0 and g["gather_ms"] > 0 and 0 < g["gather_share"] < 1
    # state + next_state + next_agent_state float32 [171] + action int64 + env_action int32 [4] + reward f32 + terminal + env_id / tick / agent int32
    assert g["bytes_all_ranks"] == g["experiences"] * (3 * 171 * 4 + 8 + 16 + 4 + 1 + 12)
