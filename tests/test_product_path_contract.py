"""The product path has no CPU fallback and never touches the oracle: it fails loudly without the HIP extension or without a GPU
(the tier's rule: only tests/, smoke() and bench.py's cpu_baseline leg may use anything under oracle/)."""
import os
import re
import subprocess
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_missing_extension_is_an_error_not_a_fallback(tmp_path):
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from maro_amd import _lib\n"
            "from maro_amd.cim.engine import CimBatchEngine\n"
            "from maro_amd.citi_bike.engine import CitiBikeBatchEngine\n"
            "for cls, args in ((CimBatchEngine, ('toy.5p_ssddd_l0.5', 2)), (CitiBikeBatchEngine, ('toy.3s_4t', 2))):\n"
            "    try:\n"
            "        cls(*args)\n"
            "    except _lib.ExtensionMissingError as e:\n"
            "        assert 'no CPU fallback' in str(e)\n"
            "    else:\n"
            "        raise SystemExit('engine was created without the HIP extension')\n"
            "print('ok')\n" % REPO)
    env = dict(os.environ, MARO_AMD_LIB=str(tmp_path / "not_built" / "libmaro_amd.so"))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr


@pytest.mark.skipif(torch.cuda.is_available(), reason="needs a box without a GPU")
def test_no_gpu_is_an_error_not_a_fallback():
    from maro_amd.cim.engine import CimBatchEngine
    from maro_amd.citi_bike.engine import CitiBikeBatchEngine
    for cls, args in ((CimBatchEngine, ("toy.5p_ssddd_l0.5", 2)), (CitiBikeBatchEngine, ("toy.3s_4t", 2))):
        with pytest.raises(RuntimeError, match="no CPU path"):
            cls(*args)


def test_the_package_never_reaches_for_the_oracle_or_the_emulator():
    py = re.compile(r"^\s*(from|import)\s+(oracle|tests)\b|[\"']oracle[/\"']|libcim_oracle|cb_emu|cim_emu")
    inc = re.compile(r"^\s*#\s*include\s+[<\"].*(oracle|tests)/")
    hits = []
    for root, _, files in os.walk(os.path.join(REPO, "maro_amd")):
        for f in files:
            pat = py if f.endswith(".py") else inc if f.endswith((".h", ".hip", ".cpp")) else None
            if pat is None:
                continue
            for i, line in enumerate(open(os.path.join(root, f), errors="replace"), 1):
                if pat.search(line) and not line.lstrip().startswith("#") or (pat is inc and pat.search(line)):
                    hits.append(f"{os.path.relpath(os.path.join(root, f), REPO)}:{i}: {line.strip()}")
    assert not hits, hits


def test_the_exchange_functions_hold_only_the_path_a_multi_gpu_run_takes():
    """maro_amd/cim/rollout.py posts its tensors to torch.distributed as they are: no branch on the process group's backend and no
    host staging copy (a harness that needs one injects tests/transport.py::HostStaging)."""
    src = open(os.path.join(REPO, "maro_amd", "cim", "rollout.py")).read()
    code = "\n".join(line.split("#")[0] for line in src.splitlines())
    code = re.sub(r'"""(?s:.*?)"""', "", code)
    assert "get_backend" not in code and ".cpu()" not in code
    from maro_amd.cim.rollout import Transport
    t = torch.arange(6).view(2, 3)
    tp = Transport()
    assert tp.outbound(t) is t and tp.inbound(t, t.device) is t
    land = tp.landing([4, 3], t)
    assert land.shape == (4, 3) and land.dtype == t.dtype and land.device == t.device
