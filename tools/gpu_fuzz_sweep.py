"""One-off wider differential sweep on the GPU: random CIM topologies and citi_bike data sets, HIP engine vs the oracle
(the same run_case the tests use, more seeds).  usage: python tools/gpu_fuzz_sweep.py <first_seed> <count>"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402

from tests.cb_gpu_backend import CbGpuBackend  # noqa: E402
from tests.fuzz_citi_bike import run_case as cb_case  # noqa: E402
from tests.fuzz_topologies import run_case as cim_case  # noqa: E402
from tests.gpu_backend import GpuBackend  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
t0, bad = time.time(), []
for s in range(first, first + count):
    for name, fn, be in (("cim", cim_case, GpuBackend), ("citi_bike", cb_case, CbGpuBackend)):
        try:
            fn(s, backend=be)
        except Exception as e:  # noqa: BLE001
            bad.append((name, s, repr(e)[:200]))
from maro_amd.cim import specialize  # noqa: E402
print(f"{count} seeds x 2 scenarios in {time.time() - t0:.0f} s; failures: {len(bad)}; engines that ran plan-specialised kernels "
      f"(MARO_AMD_SPECIALIZE={os.environ.get('MARO_AMD_SPECIALIZE', '0')}): {specialize.LOADS}")
for b in bad[:20]:
    print(b)
