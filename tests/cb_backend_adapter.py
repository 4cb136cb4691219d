"""Gives a citi_bike batch backend (CPU harness or GPU engine) the single-env surface of CitiBikeOracle, so the
same golden replays drive the oracle, the host-compiled device code and the HIP engine."""
import numpy as np

from maro_amd.citi_bike.abi import NODE_TYPE, STATION_ATTRS

FL_FRESH, FL_FINISHED = 1, 2


class CbBackendEnv:
    def __init__(self, backend, env=0):
        self.b, self.e = backend, env
        self.S = backend.data.n_stations
        self._done = False

    def step(self, actions=None):
        b = self.b
        a = np.full((b.n_envs, b.max_actions, 3), -1, np.int32)
        na = np.zeros(b.n_envs, np.int32)
        if actions:
            for i, act in enumerate(actions):
                a[:, i] = act
            na[:] = len(actions)
        if self._done:
            return None, None, True
        dec, scope, met, done = b.step(a, na)
        for x in (dec, scope, met, done):  # identical envs must stay identical
            assert (x == x[self.e:self.e + 1]).all()
        d, s, m = dec[self.e], scope[self.e], met[self.e]
        self._last = d
        metrics = dict(trip_requirements=int(m[0]), bike_shortage=int(m[1]), operation_number=int(m[2]))
        if done[self.e]:
            self._done = True
            return metrics, None, True
        return metrics, dict(tick=int(d[0]), station_idx=int(d[1]), type=int(d[2]), frame_index=int(d[3]),
                             action_scope=[(int(s[i, 0]), int(s[i, 1])) for i in range(d[4])]), False

    def step_joint(self, actions_per_event=None):
        """Joint modes: actions_per_event[i] = action list (or None) of the i-th reported event."""
        b = self.b
        S = self.S
        a = np.full((b.n_envs, S, b.max_actions, 3), -1, np.int32)
        na = np.zeros((b.n_envs, S), np.int32)
        nans = np.zeros(b.n_envs, np.int32)
        for i, acts in enumerate(actions_per_event or []):
            for j, act in enumerate(acts or []):
                a[:, i, j] = act
            na[:, i] = len(acts or [])
        nans[:] = len(actions_per_event or [])
        if self._done:
            return None, None, True
        dec, scope, met, done = b.step_joint(a, na, nans)
        for x in (dec, scope, met, done):
            assert (x == x[self.e:self.e + 1]).all()
        d, s, m = dec[self.e], scope[self.e], met[self.e]
        metrics = dict(trip_requirements=int(m[0]), bike_shortage=int(m[1]), operation_number=int(m[2]))
        if done[self.e]:
            self._done = True
            return metrics, None, True
        n_ev = int(d[0, 6])
        assert n_ev >= 1 and all(d[k, 5] == 1 and d[k, 7] == k for k in range(n_ev)) and (n_ev == S or d[n_ev, 5] == 0)
        return metrics, [dict(tick=int(d[k, 0]), station_idx=int(d[k, 1]), type=int(d[k, 2]), frame_index=int(d[k, 3]),
                              action_scope=[(int(s[k, i, 0]), int(s[k, i, 1])) for i in range(d[k, 4])]) for k in range(n_ev)], False

    @property
    def tick(self):
        return int(self.b.hdr()[0, self.e])

    def frame_indices(self):
        hdr = self.b.hdr()[:, self.e]
        fis = set(int(x) for x in self.b.ring_fi()[:, self.e] if x >= 0)
        if not (hdr[1] & (FL_FRESH | FL_FINISHED)):
            cur = (int(hdr[0]) - self.b.start_tick) // self.b.res
            slots = self.b.layout.ring_slots
            fis = {f for f in fis if f % slots != cur % slots} | {cur}
        return sorted(fis)

    def query(self, node, ticks, nodes, attrs):
        ticks = list(ticks) if len(ticks) else self.frame_indices()
        if node == "matrices":
            out = self.b.query(NODE_TYPE[node], ticks, [0], [0], self.S * self.S)
        else:
            nodes = list(nodes) if len(nodes) else list(range(self.S))
            out = self.b.query(NODE_TYPE[node], ticks, nodes, [STATION_ATTRS.index(a) for a in attrs], len(attrs))
        return out[self.e].reshape(-1)
