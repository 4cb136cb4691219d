"""Single-env, oracle-shaped adapter over a batch backend (CPU emulation or the HIP engine), so the
golden replay in tests/test_oracle_golden.py can drive any of them."""
import numpy as np

from oracle.cim_oracle import NODE_ATTRS, NODE_TYPE


class SingleEnvAdapter:
    def __init__(self, backend, env=0, seed=None):
        self.b, self.e = backend, env
        self.topo = backend.topo
        self._pending_seed = None
        self.lay = backend.layout
        cmd = np.full(backend.n_envs, self.topo.seed if seed is None else seed, np.int64)
        backend.reset(cmd)
        self._paused = False

    # ---- reference-shaped control (core.py:143-170, 219-229; cim_data_container_helpers.py:56-73)
    def set_seed(self, s):
        self._pending_seed = int(s)

    def reset(self, keep_seed=False):
        if not keep_seed:
            cmd = -2
        elif self._pending_seed is not None:
            cmd = self._pending_seed
        else:
            cmd = -1
        self._pending_seed = None
        self._paused = False
        self.b.reset(np.full(self.b.n_envs, cmd, np.int64))

    def step(self, actions=None):
        A = self.b.max_actions
        acts = np.full((self.b.n_envs, A, 4), -1, np.int32)
        na = np.zeros(self.b.n_envs, np.int32)
        if actions:
            assert len(actions) <= A
            for i, a in enumerate(actions):
                acts[:, i] = a
            na[:] = len(actions)
        dec, met, done = self.b.step(acts, na)
        self._paused = int(dec[self.e][7]) == 1
        self._cur_fi = int(dec[self.e][6])
        return met[self.e], dec[self.e], bool(done[self.e])

    # ---- introspection
    def _v(self, off, dtype, shape):
        return self.b.view(off, dtype, shape)

    @property
    def tick(self):
        return int(self._v(self.lay.off_tick, np.int32, (self.b.n_envs,))[self.e])

    @property
    def error(self):
        return int(self._v(self.lay.off_status, np.int32, (self.b.n_envs,))[self.e])

    @property
    def data_seed(self):
        return int(self._v(self.lay.off_seed, np.int64, (self.b.n_envs,))[self.e])

    def stops(self, v):
        V, S = self.lay.n_vessels, self.lay.max_stops
        n = int(self._v(self.lay.off_nstops, np.int32, (self.b.n_envs, V))[self.e, v])
        packed = self._v(self.lay.off_stops, np.uint32, (self.b.n_envs, V, S))[self.e, v, :n]
        arr = (packed >> 8).astype(np.int32)
        leave = arr + (packed & 0xFF).astype(np.int32)
        t = self.topo
        r = int(t.vessel_route[v])
        L = int(t.route_offset[r + 1] - t.route_offset[r])
        port = np.array([t.route_port[t.route_offset[r] + (int(t.vessel_start_offset[v]) + k) % L] for k in range(n)],
                        np.int32)
        return arr, leave, port

    def order_proportion(self):
        return self._v(self.lay.off_order_prop, np.int32, (self.b.n_envs, self.b.max_tick))[self.e].copy()

    def vessel_period(self):
        return self._v(self.lay.off_vessel_period, np.int32, (self.b.n_envs, self.lay.n_vessels))[self.e].copy()

    def frame_indices(self):
        fi = self._v(self.lay.off_ring_fi, np.int32, (self.b.n_envs, self.lay.ring_slots))[self.e]
        S = self.lay.ring_slots
        out = [int(x) for i, x in enumerate(fi) if x >= 0 and not (self._paused and i == self._cur_fi % S)]
        if self._paused:  # the current frame of a paused env is its live frame (aliased pre-decision snapshot)
            out.append(self._cur_fi)
        return sorted(out)

    def _row_slots(self, node, attrs):
        t = self.topo
        n = 0
        for a in attrs:
            if node == "vessels" and a in ("past_stop_list", "past_stop_tick_list"):
                n += t.past_stop_number
            elif node == "vessels" and a in ("future_stop_list", "future_stop_tick_list"):
                n += t.future_stop_number
            elif node == "matrices":
                n += t.n_ports * t.n_ports if a == "full_on_ports" else t.n_vessels * t.n_ports
            else:
                n += 1
        return n

    def query(self, node, ticks, nodes, attrs):
        t = self.topo
        ticks = list(ticks) if len(ticks) else self.frame_indices()
        n_nodes = {"ports": t.n_ports, "vessels": t.n_vessels, "matrices": 1}[node]
        nodes = list(nodes) if len(nodes) else list(range(n_nodes))
        ids = [NODE_ATTRS[node].index(a) for a in attrs]
        if not ticks:
            return np.zeros(0, np.float64)
        out = self.b.query(NODE_TYPE[node], ticks, nodes, ids, self._row_slots(node, attrs))
        return np.asarray(out[self.e]).reshape(-1)
