"""CitiBikeBatchEngine look-alike on top of the host-compiled device code — lets the `-m "not gpu"` suite exercise the
host-side object API (maro_amd/citi_bike/vector_env.py) without a GPU.  Test infrastructure only."""
import numpy as np
import torch

from maro_amd.citi_bike.abi import NODE_ATTRS, NODE_TYPE, draw_transfer_times
from maro_amd.citi_bike.data import load_topology
from tests.emu.cb_emu import CbEmuBackend


class CbEmuEngine:
    def __init__(self, topology, n_envs, start_tick=0, durations=1440, snapshot_resolution=1, max_snapshots=None, max_actions=1,
                 seeds=None, decision_mode=0):
        self.data = topology if not isinstance(topology, str) else load_topology(topology)
        self.b = CbEmuBackend(self.data, n_envs, start_tick, durations, snapshot_resolution, max_snapshots, max_actions, decision_mode=decision_mode)
        self.decision_mode = decision_mode
        self.n_envs, self.max_actions = n_envs, max_actions
        self.start_tick, self.durations, self.snapshot_resolution = start_tick, durations, snapshot_resolution
        self.max_tick = start_tick + durations
        self.layout = self.b.layout
        self.hdr = torch.from_numpy(self.b.hdr())
        self.ticks, self.status = self.hdr[0], self.hdr[13]
        self.ring_fi = torch.from_numpy(self.b.ring_fi())
        self.reset(seeds=np.arange(n_envs) if seeds is None else seeds)

    def reset(self, seeds=None, transfer_times=None, mask=None):
        if seeds is not None:
            transfer_times = draw_transfer_times(self.data, seeds, self.layout.transfer_times_cap)
        self.b.reset(transfer_times, None if mask is None else np.asarray(mask))

    def step(self, actions=None, n_actions=None, mask=None):
        out = self.b.step(None if actions is None else np.asarray(actions), None if n_actions is None else np.asarray(n_actions),
                          None if mask is None else np.asarray(mask))
        self.decisions, self.scope, self.metrics, self.done = (torch.from_numpy(x) for x in out)
        return self.decisions, self.scope, self.metrics, self.done

    def step_joint(self, actions=None, n_actions=None, n_answered=None, mask=None):
        out = self.b.step_joint(None if actions is None else np.asarray(actions), None if n_actions is None else np.asarray(n_actions),
                                None if n_answered is None else np.asarray(n_answered), None if mask is None else np.asarray(mask))
        self.decisions, self.scope, self.metrics, self.done = (torch.from_numpy(x) for x in out)
        return self.decisions, self.scope, self.metrics, self.done

    def query(self, node, ticks, nodes, attrs, out=None):
        ids = [NODE_ATTRS[node].index(a) for a in attrs]
        slots = sum(self.data.n_stations ** 2 if node == "matrices" else 1 for _ in attrs)
        return torch.from_numpy(self.b.query(NODE_TYPE[node], np.asarray(ticks, np.int32), np.asarray(nodes, np.int32), ids, slots))
