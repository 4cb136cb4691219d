"""ORACLE for the citi_bike scenario — test infrastructure only (pure Python; small toy topologies).

An event-driven restatement of the reference path, keeping its structure (per-tick event lists executed in
insertion order, events appended to the running tick, lazily evaluated action scope with the trip-window cache):
    Env loop                       maro/simulator/core.py:317-381
    CitibikeBusinessEngine         maro/simulator/scenarios/citi_bike/business_engine.py:101-147, 370-559
    BikeDecisionStrategy + filters maro/simulator/scenarios/citi_bike/decision_strategy.py:11-391
    Station callbacks              maro/simulator/scenarios/citi_bike/station.py:58-75
    snapshot ring                  maro/backends/np_backend.pyx:481-549
Pinned against vectors produced by the real reference (oracle/gen_golden_citi_bike.py) in
tests/test_citi_bike_oracle.py.  Transfer times (`round(np.random.normal(mean, std))`, global numpy RNG in the
reference, decision_strategy.py:213-216) are an explicit pre-drawn sequence here.
"""
from __future__ import annotations

from math import floor
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

STATION_ATTRS = ["bikes", "shortage", "trip_requirement", "fulfillment", "capacity", "id", "weekday", "temperature",
                 "weather", "holiday", "extra_cost", "transfer_cost", "failed_return", "min_bikes"]
SUPPLY, DEMAND = 0, 1
EV_RETURN, EV_DELIVER, EV_REQUIRE, EV_REBALANCE, EV_DECISION = range(5)


def draw_transfer_times(data, seed: int, count: int) -> np.ndarray:
    """What `np.random.seed(seed)` followed by `count` reads of BikeDecisionStrategy.transfer_time yields."""
    rs = np.random.RandomState(seed)
    return np.array([round(rs.normal(data.time_mean, scale=data.time_std)) for _ in range(count)], np.int32)


class CitiBikeOracle:
    def __init__(self, data, start_tick: int = 0, durations: int = 100, snapshot_resolution: int = 1,
                 max_snapshots: Optional[int] = None, transfer_times: Optional[Sequence[int]] = None):
        self.d = data
        self.S = data.n_stations
        self.start_tick, self.max_tick, self.res = start_tick, start_tick + durations, snapshot_resolution
        total = -(-durations // snapshot_resolution)
        self.ring_size = max_snapshots or total
        nb = data.neighbors()   # once: it sorts every station's row
        self.neighbors = [[(int(i), float(data.distance[s][i])) for i in nb[s] if i >= 0] for s in range(self.S)]
        self.trip_off = data.trip_offsets(self.max_tick)
        self._transfer_times = list(transfer_times) if transfer_times is not None else []
        self.reset()

    # ------------------------------------------------------------------ state
    def reset(self):
        S = self.S
        z = lambda: np.zeros(S, np.int64)  # noqa: E731
        self.bikes, self.capacity = self.d.init_bikes.astype(np.int64).copy(), self.d.capacity.astype(np.int64).copy()
        self.min_bikes = self.bikes.copy()
        self.shortage, self.trip_requirement, self.fulfillment = z(), z(), z()
        self.extra_cost, self.transfer_cost, self.failed_return = z(), z(), z()
        self.weekday, self.holiday, self.weather, self.temperature = z(), z(), z(), z()
        self.trips_adj = np.zeros((S, S), np.int64)
        self.total_trips = self.total_shortages = self.total_operate = 0
        self.events: Dict[int, list] = {}
        self.tick = self.start_tick
        self.last_day = None
        self.snap: Dict[int, dict] = {}      # frame index -> copy of the frame
        self.snap_order: List[int] = []
        self.window_cache: Dict[int, np.ndarray] = {}
        self._tt_pos = 0
        self._fresh, self._finished, self._pending = True, False, None
        self._cursor = 0

    def _set_bikes(self, s: int, v: int):  # station.py:71-75
        self.bikes[s] = v
        self.min_bikes[s] = min(v, self.min_bikes[s])

    def _frame(self) -> dict:
        return {k: getattr(self, k).copy() for k in ("bikes", "shortage", "trip_requirement", "fulfillment", "capacity",
                                                     "weekday", "temperature", "weather", "holiday", "extra_cost",
                                                     "transfer_cost", "failed_return", "min_bikes", "trips_adj")}

    def _take_snapshot(self, fi: int):  # np_backend.pyx:481-518
        if fi not in self.snap and len(self.snap_order) >= self.ring_size:
            old = self.snap_order.pop(0)
            del self.snap[old]
        if fi in self.snap:
            self.snap_order.remove(fi)
        self.snap[fi] = self._frame()
        self.snap_order.append(fi)

    def frame_index(self, tick: int) -> int:
        return (tick - self.start_tick) // self.res

    # ------------------------------------------------------------------ events
    def _insert(self, tick: int, ev: tuple):
        self.events.setdefault(tick, []).append(ev)

    def _be_step(self, tick: int):  # business_engine.py:101-128
        for i in range(self.trip_off[tick], self.trip_off[tick + 1]):
            self._insert(tick, (EV_REQUIRE, int(self.d.trip_src[i]), int(self.d.trip_dst[i]), int(self.d.trip_duration[i])))
        if (tick + 1) % self.d.resolution == 0:
            self._insert(tick, (EV_REBALANCE,))
        day = int(self.d.tick_day[tick])  # :370-396
        if day != self.last_day:
            self.last_day = day
            self.weekday[:] = self.d.day_weekday[day]
            self.holiday[:] = self.d.day_holiday[day]
            self.weather[:] = self.d.day_weather[day]
            self.temperature[:] = self.d.day_temperature[day]

    def _move_to_neighbor(self, src: int, cur: int, number: int):  # decision_strategy.py:295-343
        for order_index, (nb, _dist) in enumerate(self.neighbors[cur]):
            accept = min(int(self.capacity[nb] - self.bikes[nb]), number)
            self._set_bikes(nb, int(self.bikes[nb]) + accept)
            cost = accept * (order_index + 1)
            target = src if self.d.extra_cost_mode == 0 else cur if self.d.extra_cost_mode == 1 else nb
            self.extra_cost[target] += cost
            number -= accept
            if number == 0:
                break

    def _execute(self, tick: int):
        """event_buffer.execute: run events of `tick` from the cursor; stop at the first pending decision."""
        lst = self.events.get(tick, [])
        while self._cursor < len(lst):
            ev = lst[self._cursor]
            if ev[0] == EV_DECISION:
                return ev
            self._cursor += 1
            if ev[0] == EV_REQUIRE:  # :398-437
                _, src, dst, dur = ev
                b = int(self.bikes[src])
                self.trip_requirement[src] += 1
                self.total_trips += 1
                self.trips_adj[src, dst] += 1
                if b < 1:
                    self.shortage[src] += 1
                    self.total_shortages += 1
                else:
                    self.fulfillment[src] += 1
                    self._set_bikes(src, b - 1)
                    self._insert(tick + dur, (EV_RETURN, src, dst, 1))
                    lst = self.events.get(tick, [])
            elif ev[0] in (EV_RETURN, EV_DELIVER):  # :439-466, :494-519
                _, frm, to, n = ev
                b = int(self.bikes[to])
                accept = min(int(self.capacity[to]) - b, n)
                if accept < n:
                    if ev[0] == EV_RETURN:
                        self.failed_return[to] += n - accept
                    self._move_to_neighbor(frm, to, n - accept)
                if ev[0] == EV_DELIVER and accept > 0:
                    self.transfer_cost[to] += accept
                    self.total_operate += accept
                self._set_bikes(to, b + accept)
            elif ev[0] == EV_REBALANCE:  # :468-492 + decision_strategy.py:229-251
                for s in range(self.S):
                    ratio = int(self.bikes[s]) / int(self.capacity[s])
                    if ratio >= self.d.supply_water_mark_ratio:
                        self._insert(tick, (EV_DECISION, s, SUPPLY))
                    elif ratio <= self.d.demand_water_mark_ratio:
                        self._insert(tick, (EV_DECISION, s, DEMAND))
                lst = self.events.get(tick, [])
        return None

    # ------------------------------------------------------------------ action scope (decision_strategy.py:253-293)
    def _action_scope(self, s: int, dtype: int) -> List[Tuple[int, int]]:
        scope: Dict[int, int] = {}
        for nb, _ in self.neighbors[s]:
            scope[nb] = int(self.capacity[nb] - self.bikes[nb]) if dtype == SUPPLY else floor(int(self.bikes[nb]) * self.d.scope_high_ratio)
        for f in self.d.filters:
            n_out = min(f["num"], len(scope))
            if f["type"] == 0:      # DistanceFilter :11-49
                scope = {nb: scope[nb] for nb, _ in self.neighbors[s][:n_out]}
            elif f["type"] == 1:    # RequirementsFilter :52-85
                items = sorted(scope.items(), key=lambda kv: (kv[1], kv[0]), reverse=True)
                scope = {items[i][0]: items[i][1] for i in range(n_out)}
            else:                   # TripsWindowFilter :88-163 (caches frames the first time it sees them)
                fis = list(self.snap_order)
                windows = min(f["windows"], len(fis))
                fis = fis[-windows:]   # NB `lst[-0:]` is the whole list: windows == 0 means every stored frame (:123-129)
                trips: Dict[int, int] = {}
                for i, fi in enumerate(fis):
                    if i == windows - 1 or fi not in self.window_cache:
                        self.window_cache[fi] = self.snap[fi]["trip_requirement"].copy()
                    st = self.window_cache[fi]
                    for nb in scope:
                        trips[nb] = trips.get(nb, 0) + int(st[nb])
                items = sorted(trips.items(), key=lambda kv: (kv[1], kv[0]), reverse=(dtype == DEMAND))
                scope = {nb: scope[nb] for nb, _ in items[:n_out]}
        scope[s] = floor(int(self.bikes[s]) * (1 - self.d.scope_low_ratio)) if dtype == SUPPLY else int(self.capacity[s] - self.bikes[s])
        return list(scope.items())

    # ------------------------------------------------------------------ Env.step, Sequential mode
    def step(self, actions: Optional[Sequence[Tuple[int, int, int]]] = None):
        """actions: [(from_station_idx, to_station_idx, number)].  Returns (metrics, decision | None, done) where
        decision = dict(tick, station_idx, type, frame_index, action_scope=[(station, max)...])."""
        if self._finished:
            return None, None, True
        if self._pending is not None:
            # TAKE_ACTION is an immediate event of the decision: it runs right after it (:521-559)
            self._cursor += 1
            # Reference quirk (event_linked_list.py:86-108): popping a finished cascade event that was the LAST element
            # of the tick's list leaves `_tail` pointing at it, so events appended to this same tick afterwards
            # (a DeliverBike with transfer time 0) hang off the removed node and are never executed.
            tail_stale = self._cursor == len(self.events.get(self.tick, []))
            for frm, to, number in (actions or []):
                if frm < 0 or to < 0:
                    continue
                b = int(self.bikes[frm])
                ex = min(b, int(number))
                if ex > 0:
                    self._set_bikes(frm, b - ex)
                    tt = int(self._transfer_times[self._tt_pos])
                    self._tt_pos += 1
                    if not (tt == 0 and tail_stale):
                        self._insert(self.tick + tt, (EV_DELIVER, frm, to, ex))
            self._pending = None
        elif self._fresh:
            self._fresh = False
            self._cursor = 0
            self._be_step(self.tick)
        while True:
            ev = self._execute(self.tick)
            if ev is not None:
                fi = self.frame_index(self.tick)
                self._take_snapshot(fi)  # core.py:345
                _, s, dtype = ev
                self._pending = ev
                return self.metrics(), dict(tick=self.tick, station_idx=s, type=dtype, frame_index=fi,
                                            action_scope=self._action_scope(s, dtype)), False
            # post_step :130-147
            if (self.tick + 1) % self.res == 0:
                self._take_snapshot(self.frame_index(self.tick))
                for a in ("shortage", "trip_requirement", "extra_cost", "transfer_cost", "fulfillment", "failed_return"):
                    getattr(self, a)[:] = 0
                self.min_bikes[:] = self.bikes
            if self.tick + 1 == self.max_tick:
                break
            self.events.pop(self.tick, None)
            self.tick += 1
            self._cursor = 0
            self._be_step(self.tick)
        if (self.tick + 1) % self.res != 0:
            self._take_snapshot(self.frame_index(self.tick))
        self._finished = True
        return self.metrics(), None, True

    # ------------------------------------------------------------------ Env.step, Joint modes (core.py:354-366)
    def step_joint(self, actions_per_event: Optional[Sequence] = None, mode: int = 1):
        """Joint (mode 1) / JointWithSequentialAction (mode 2): every pending decision event of the tick is reported at
        once.  `actions_per_event[i]` = the action list ([(from, to, number), ...] or None) for the i-th reported event;
        events beyond len(actions_per_event) are finished (mode 1) or stay pending and are reported again (mode 2).
        Returns (metrics, [decision dicts] | None, done).  The scopes are evaluated on the state at report time (the
        reference caches a payload's scope at its first read; the object API on top re-serves it)."""
        if self._finished:
            return None, None, True
        if self._pending is not None:
            lst = self.events.get(self.tick, [])
            pend = self._pending
            acts = list(actions_per_event or [])[: len(pend)]
            for i, a in enumerate(acts):
                # the decision event runs (no handler), is popped, and its TAKE_ACTION event runs right behind it
                self._cursor += 1
                tail_stale = self._cursor == len(self.events.get(self.tick, []))   # event_linked_list.py:86-108, see step()
                for frm, to, number in (a or []):
                    if frm < 0 or to < 0:
                        continue
                    b = int(self.bikes[frm])
                    ex = min(b, int(number))
                    if ex > 0:
                        self._set_bikes(frm, b - ex)
                        tt = int(self._transfer_times[self._tt_pos])
                        self._tt_pos += 1
                        if not (tt == 0 and tail_stale):
                            self._insert(self.tick + tt, (EV_DELIVER, frm, to, ex))
            if mode == 1:
                self._cursor += len(pend) - len(acts)   # unanswered events: state FINISHED, popped without effect
            self._pending = None
            del lst
        elif self._fresh:
            self._fresh = False
            self._cursor = 0
            self._be_step(self.tick)
        while True:
            ev = self._execute(self.tick)
            if ev is not None:
                fi = self.frame_index(self.tick)
                self._take_snapshot(fi)  # core.py:345
                lst = self.events[self.tick]
                pend = []
                k = self._cursor
                while k < len(lst) and lst[k][0] == EV_DECISION:   # event_linked_list.py:113-120: consecutive decision events
                    pend.append(lst[k])
                    k += 1
                self._pending = pend
                return self.metrics(), [dict(tick=self.tick, station_idx=s, type=dtype, frame_index=fi, action_scope=self._action_scope(s, dtype))
                                        for _, s, dtype in pend], False
            if (self.tick + 1) % self.res == 0:
                self._take_snapshot(self.frame_index(self.tick))
                for a in ("shortage", "trip_requirement", "extra_cost", "transfer_cost", "fulfillment", "failed_return"):
                    getattr(self, a)[:] = 0
                self.min_bikes[:] = self.bikes
            if self.tick + 1 == self.max_tick:
                break
            self.events.pop(self.tick, None)
            self.tick += 1
            self._cursor = 0
            self._be_step(self.tick)
        if (self.tick + 1) % self.res != 0:
            self._take_snapshot(self.frame_index(self.tick))
        self._finished = True
        return self.metrics(), None, True

    def metrics(self):
        return dict(trip_requirements=self.total_trips, bike_shortage=self.total_shortages, operation_number=self.total_operate)

    # ------------------------------------------------------------------ snapshot_list[...] (np_backend.pyx:520-549)
    def frame_indices(self) -> List[int]:
        return list(self.snap_order)

    def query(self, node: str, ticks: Sequence[int], nodes: Sequence[int], attrs: Sequence[str]) -> np.ndarray:
        ticks = list(ticks) if len(ticks) else list(self.snap_order)
        out = []
        if node == "matrices":
            for fi in ticks:
                out.append(self.snap[fi]["trips_adj"].reshape(-1).astype(np.float64) if fi in self.snap else np.zeros(self.S * self.S))
            return np.concatenate(out) if out else np.zeros(0)
        nodes = list(nodes) if len(nodes) else list(range(self.S))
        for fi in ticks:
            for n in nodes:
                for a in attrs:
                    if fi not in self.snap:
                        out.append(0.0)
                    elif a == "id":
                        out.append(float(self.d.station_id[n]))
                    else:
                        out.append(float(self.snap[fi][a][n]))
        return np.array(out, np.float64)
