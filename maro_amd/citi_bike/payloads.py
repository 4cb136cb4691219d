"""citi_bike payload types, source-compatible with ``maro/simulator/scenarios/citi_bike/common.py:68-152``.

If the real ``maro`` package is importable its classes are re-exported unchanged; otherwise equivalent classes with
the same constructor signatures, attributes, pickling protocol and repr are defined here.
"""
from __future__ import annotations

from enum import Enum

try:  # pragma: no cover - depends on the host environment
    from maro.simulator.scenarios.citi_bike.common import Action, DecisionEvent, DecisionType, ExtraCostMode  # noqa: F401

    HAVE_MARO = True
except Exception:  # maro (or its `holidays` dependency) is not installed, e.g. on the GPU box
    HAVE_MARO = False

    class DecisionType(Enum):
        """common.py:57-65: Supply = the station has too many bikes, Demand = too few."""

        Supply = "supply"
        Demand = "demand"

    class ExtraCostMode(Enum):
        Source = "source"
        Target = "target"

    class Action:
        """common.py:127-152: move `number` bikes from one station to another."""

        summary_key = ["from_station_idx", "to_station_idx", "number"]

        def __init__(self, from_station_idx: int, to_station_idx: int, number: int):
            self.from_station_idx = from_station_idx
            self.to_station_idx = to_station_idx
            self.number = number

        def __repr__(self):
            return "%s {from_station_idx: %r, to_station_idx: %r, number:%r}" % (
                self.__class__.__name__, self.from_station_idx, str(self.to_station_idx), self.number)

    class DecisionEvent:
        """common.py:68-124 (same constructor: the action scope is a callable evaluated lazily)."""

        summary_key = ["station_idx", "tick", "frame_index", "type", "action_scope"]

        def __init__(self, station_idx: int, tick: int, frame_index: int, action_scope_func, decision_type: DecisionType):
            self.station_idx = station_idx
            self.tick = tick
            self.frame_index = frame_index
            self.type = decision_type
            self._action_scope = None
            self._action_scope_func = action_scope_func

        @property
        def action_scope(self) -> dict:
            if self._action_scope is None:
                self._action_scope = self._action_scope_func(self.station_idx, self.type)
            return self._action_scope

        def __getstate__(self):
            return {"station_idx": self.station_idx, "tick": self.tick, "frame_index": self.frame_index, "type": self.type,
                    "action_scope": self.action_scope}

        def __setstate__(self, state):
            self.station_idx = state["station_idx"]
            self.tick = state["tick"]
            self.frame_index = state["frame_index"]
            self.type = state["type"]
            self._action_scope = state["action_scope"]

        def __repr__(self):
            return "%s {station_idx: %r, type: %r, action_scope:%r}" % (
                self.__class__.__name__, self.station_idx, str(self.type), self.action_scope)


def encode_action(a) -> tuple:
    """Action object -> the C ABI's (from_station_idx, to_station_idx, number)."""
    return (int(a.from_station_idx), int(a.to_station_idx), int(a.number))


def make_decision_event(row, scope_rows) -> "DecisionEvent":
    """Decision + scope rows of the C ABI -> DecisionEvent.  The engine evaluates the scope at the pause (that is
    also when the reference's TripsWindowFilter cache is fed); the dict keeps the reference's insertion order."""
    tick, station, dtype, frame_index, n = (int(x) for x in row[:5])
    scope = {int(scope_rows[i][0]): int(scope_rows[i][1]) for i in range(n)}
    return DecisionEvent(station, tick, frame_index, lambda s, t: scope, DecisionType.Supply if dtype == 0 else DecisionType.Demand)
