"""numpy-facing wrapper over maro_amd.cim.engine.CimBatchEngine with the same surface as
tests/emu/emu.py::EmuBackend, so the golden replay drives the real HIP kernels through the C ABI."""
import numpy as np
import torch

from maro_amd.cim.engine import CimBatchEngine


class GpuBackend:
    def __init__(self, topo, n_envs=1, start_tick=0, durations=100, snapshot_resolution=1, max_snapshots=None,
                 max_actions=2, decision_mode=0, order_table=0):
        self.eng = CimBatchEngine(topo, n_envs, start_tick=start_tick, durations=durations,
                                  snapshot_resolution=snapshot_resolution, max_snapshots=max_snapshots,
                                  max_actions=max_actions, decision_mode=decision_mode, order_table=order_table)
        self.topo = self.eng.topo
        self.layout = self.eng.layout
        self.n_envs, self.max_actions, self.max_tick = n_envs, max_actions, start_tick + durations

    def view(self, off, dtype, shape):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        torch.cuda.synchronize()
        return self.eng.workspace[off:off + n].cpu().numpy().view(dtype).reshape(shape)

    def reset(self, seed_cmd=None, mask=None):
        self.eng.reset(seed_cmd, mask)

    def step(self, actions=None, n_actions=None, mask=None, n_answered=None):
        d, m, dn = self.eng.step(actions, n_actions, mask, n_answered=n_answered)
        torch.cuda.synchronize()
        return d.cpu().numpy(), m.cpu().numpy(), dn.cpu().numpy()

    def query(self, node_type, ticks, nodes, attrs, row_slots):
        node = ["ports", "vessels", "matrices"][node_type]
        from maro_amd.cim.engine import NODE_ATTRS
        out = self.eng.query(node, np.asarray(ticks, np.int32), np.asarray(nodes, np.int32),
                             [NODE_ATTRS[node][a] for a in attrs])
        torch.cuda.synchronize()
        return out.cpu().numpy()


    def set_observation(self, port_attr_ids, vessel_attr_ids):
        from maro_amd.cim.engine import PORT_ATTRS, VESSEL_ATTRS
        op, ov = self.eng.set_observation([PORT_ATTRS[a] for a in port_attr_ids], [VESSEL_ATTRS[a] for a in vessel_attr_ids])
        return _Live(op), _Live(ov)


class _Live:
    """np.array(x) gives the current content of a device tensor."""

    def __init__(self, t):
        self.t = t

    def __array__(self, dtype=None, copy=None):
        torch.cuda.synchronize()
        return self.t.cpu().numpy()
