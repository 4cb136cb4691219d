import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from maro_amd.cim import specialize as spec
from maro_amd.citi_bike.abi import MrxCbConfig, topology_struct
from maro_amd.citi_bike.data import load_topology as load_cb
data = load_cb("toy.3s_4t")
ts, keep = topology_struct(data)
cap = data.n_stations * (int((data.time_mean + 6 * data.time_std) / max(data.resolution, 1)) + 2) + 4
for n in (4096,):
    d = spec.plan_defines(ts, MrxCbConfig(n, 0, 0, 44000, 10, 16, 1, cap, 0), "citi_bike")
    print(len(spec.code_object(d, scenario="citi_bike")), spec.CSRC, spec.CACHE, spec.FLAGS[-1])
