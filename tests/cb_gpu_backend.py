"""numpy-facing wrapper over maro_amd.citi_bike.engine.CitiBikeBatchEngine with the surface of
tests/emu/cb_emu.py::CbEmuBackend, so the same replays drive the real HIP kernels through the C ABI."""
import numpy as np
import torch

from maro_amd.citi_bike.abi import NODE_ATTRS
from maro_amd.citi_bike.engine import CitiBikeBatchEngine


class CbGpuBackend:
    def __init__(self, data, n_envs=1, start_tick=0, durations=100, snapshot_resolution=1, max_snapshots=None, max_actions=1,
                 delivery_capacity=0, transfer_times_cap=0, decision_mode=0, specialize=None):
        self.eng = CitiBikeBatchEngine(data, n_envs, start_tick=start_tick, durations=durations, snapshot_resolution=snapshot_resolution,
                                       max_snapshots=max_snapshots, max_actions=max_actions, delivery_capacity=delivery_capacity,
                                       transfer_times_cap=transfer_times_cap, decision_mode=decision_mode, specialize=specialize)
        self.data, self.layout = data, self.eng.layout
        self.n_envs, self.max_actions = n_envs, max_actions
        self.start_tick, self.max_tick, self.res = start_tick, start_tick + durations, snapshot_resolution

    def reset(self, transfer_times=None, mask=None):
        tt = None if transfer_times is None else np.ascontiguousarray(transfer_times, np.int32).reshape(self.n_envs, -1)
        self.eng.reset(transfer_times=tt, mask=mask)

    def set_step_budget(self, max_records):
        self.eng.set_step_budget(max_records)

    def step(self, actions=None, n_actions=None, mask=None):
        out = self.eng.step(actions, n_actions, mask)
        torch.cuda.synchronize()
        return tuple(x.cpu().numpy() for x in out)

    def step_joint(self, actions=None, n_actions=None, n_answered=None, mask=None):
        out = self.eng.step_joint(actions, n_actions, n_answered, mask)
        torch.cuda.synchronize()
        return tuple(x.cpu().numpy() for x in out)

    def random_policy(self, dec, scope, step):
        a = torch.zeros((self.n_envs, self.max_actions, 3), dtype=torch.int32, device=self.eng.device)
        na = torch.zeros(self.n_envs, dtype=torch.int32, device=self.eng.device)
        self.eng.random_policy(step, a, na)
        torch.cuda.synchronize()
        return a.cpu().numpy(), na.cpu().numpy()

    def query(self, node_type, ticks, nodes, attrs, row_slots):
        node = ["stations", "matrices"][node_type]
        out = self.eng.query(node, np.asarray(ticks, np.int32), np.asarray(nodes, np.int32), [NODE_ATTRS[node][a] for a in attrs])
        torch.cuda.synchronize()
        return out.cpu().numpy()

    def hdr(self):
        torch.cuda.synchronize()
        return self.eng.hdr.cpu().numpy()

    def ring_fi(self):
        torch.cuda.synchronize()
        return self.eng.ring_fi.cpu().numpy()
