"""maro_amd — MI355X-native batched rollout engines for MARO's CIM and citi_bike simulators.

    from maro_amd import CimBatchEngine, CitiBikeBatchEngine, GpuVectorEnv

The compute path is the HIP library maro_amd/csrc/libmaro_amd.so (C ABI: include/maro_amd.h, include/maro_amd_citi_bike.h); there is no
CPU fallback.  Importing this package does not load the library; constructing an engine does.
"""
__version__ = "0.1.0"


def __getattr__(name):
    if name == "CimBatchEngine":
        from .cim.engine import CimBatchEngine
        return CimBatchEngine
    if name in ("GpuVectorEnv", "GpuEnvView"):
        from .cim import vector_env
        return getattr(vector_env, name)
    if name in ("Action", "ActionType", "ActionScope", "DecisionEvent"):
        from .cim import payloads
        return getattr(payloads, name)
    if name in ("CitiBikeBatchEngine",):
        from .citi_bike.engine import CitiBikeBatchEngine
        return CitiBikeBatchEngine
    if name in ("load_topology", "CimTopology"):
        from .cim import topology
        return getattr(topology, name)
    raise AttributeError(name)
