"""The drop-in claim on the GPU: the REFERENCE'S OWN code — maro.rl's ``CIMEnvSampler`` (examples/cim/rl/env_sampler.py over
maro/rl/rollout/env_sampler.py:438-611), ``examples/vector_env/hello.py`` and the citi_bike ``GreedyPolicy`` example — drives
``GpuVectorEnv`` / ``env_view(0)`` backed by libmaro_amd.so on cuda:0 and must see exactly what it sees on the reference ``Env``.

The reference comes from ``oracle/_ref/maro_ref.tgz`` (built by oracle/build_ref.sh in the build container, shipped with the
snapshot, unpacked by bench.reference_runtime() into a private cache folder); each check runs in a child interpreter that imports
the reference — this process never does.  Skipped only where no built reference is reachable."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _runtime():
    sys.path.insert(0, REPO)
    import bench
    rt = bench.reference_runtime()
    if rt is None:
        pytest.skip("no built reference on this box (oracle/_ref/maro_ref.tgz absent)")
    return rt


def _run(script, rt, extra, timeout=900):
    root, stubs, home, _ = rt
    env = dict(os.environ, HOME=home, MARO_ORACLE_HOME=home, SKIP_DEPLOYMENT="TRUE")
    out = subprocess.run([sys.executable, os.path.join(REPO, "oracle", script), "--maro", root, "--stubs", stubs, "--backend", "gpu"] + extra,
                         capture_output=True, text=True, timeout=timeout, env=env, cwd=REPO)
    assert out.returncode == 0 and "OK [gpu]" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])
    return out.stdout


@pytest.mark.parametrize("topology,durations", [("toy.5p_ssddd_l0.5", 150), ("global_trade.22p_l0.8", 120)])
def test_the_references_env_sampler_collects_the_same_experiences_on_the_hip_engine(topology, durations):
    out = _run("check_env_sampler_dropin.py", _runtime(), ["--topology", topology, "--durations", str(durations)])
    assert "experience elements" in out


def test_the_references_vector_env_example_runs_unchanged_on_the_hip_engine():
    _run("check_vector_env_example_dropin.py", _runtime(), [])


@pytest.mark.parametrize("topology", ["toy.5s_6t", "toy.3s_4t"])
def test_the_references_citi_bike_greedy_agent_sees_the_same_episode_on_the_hip_engine(topology):
    rt = _runtime()
    from oracle.setup_toy_topologies import ensure_toy
    ensure_toy(rt[0], rt[2], topology)       # the toy's build folder + config.yml written back from the packaged .npz (checker tooling)
    _run("check_citi_bike_greedy_dropin.py", rt, ["--topology", topology, "--durations", "1440"])
