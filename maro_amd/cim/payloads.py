"""CIM payload types, source-compatible with ``maro/simulator/scenarios/cim/common.py:18-150``.

If the real ``maro`` package is importable its classes are re-exported unchanged (so
``isinstance(x, maro...Action)`` keeps working inside maro.rl); otherwise equivalent classes with the
same constructor signatures, attributes, pickling protocol and repr are defined here.
"""
from __future__ import annotations

from enum import Enum

try:  # pragma: no cover - depends on the host environment
    from maro.simulator.scenarios.cim.common import Action, ActionScope, ActionType, DecisionEvent  # noqa: F401

    HAVE_MARO = True
except Exception:  # maro is not installed (e.g. the GPU box)
    HAVE_MARO = False

    class ActionType(Enum):
        """Type of CIM action (cim/common.py:18-22)."""

        LOAD = "load"
        DISCHARGE = "discharge"

    class Action:
        """cim/common.py:25-53."""

        summary_key = ["port_idx", "vessel_idx", "action_type", "quantity"]

        def __init__(self, vessel_idx: int, port_idx: int, quantity: int, action_type: ActionType):
            assert action_type is not None
            assert quantity >= 0
            self.vessel_idx = vessel_idx
            self.port_idx = port_idx
            self.quantity = quantity
            self.action_type = action_type

        def __repr__(self):
            return "%s {action_type: %r, port_idx: %r, vessel_idx: %r, quantity: %r}" % (
                self.__class__.__name__, str(self.action_type), self.port_idx, self.vessel_idx, self.quantity)

    class ActionScope:
        """cim/common.py:56-69."""

        def __init__(self, load: int, discharge: int):
            self.load = load
            self.discharge = discharge

        def __repr__(self):
            return "%s {load: %r, discharge: %r}" % (self.__class__.__name__, self.load, self.discharge)

    class DecisionEvent:
        """cim/common.py:72-150 (same constructor: scope / early discharge are callables evaluated lazily)."""

        summary_key = ["tick", "port_idx", "vessel_idx", "snapshot_list", "action_scope", "early_discharge"]
        # (class-level defaults: an event made from an engine row — _from_row, maro_amd/csrc_host/fastobj.c — carries six instance
        #  attributes instead of nine)
        _action_scope = None
        _action_scope_func = None
        _early_discharge_func = None
        _scope = None

        def __init__(self, tick, port_idx, vessel_idx, snapshot_list, action_scope_func, early_discharge_func):
            self.tick = tick
            self.port_idx = port_idx
            self.vessel_idx = vessel_idx
            self.snapshot_list = snapshot_list
            self._action_scope = None
            self._early_discharge = None
            self._action_scope_func = action_scope_func
            self._early_discharge_func = early_discharge_func

        @classmethod
        def _from_row(cls, tick, port_idx, vessel_idx, snapshot_list, load, discharge, early):
            """The engine's decision row -> event without the two closures of the reference's constructor form (one dict literal;
            the ActionScope object is made when it is read).  Same attributes, same pickled state."""
            ev = cls.__new__(cls)
            ev.__dict__ = {"tick": tick, "port_idx": port_idx, "vessel_idx": vessel_idx, "snapshot_list": snapshot_list, "_early_discharge": early,
                           "_scope": (load, discharge)}
            return ev

        @property
        def action_scope(self) -> ActionScope:
            if self._action_scope is None:
                f = self._action_scope_func
                self._action_scope = ActionScope(*self._scope) if f is None else f(self.port_idx, self.vessel_idx)
            return self._action_scope

        @property
        def early_discharge(self) -> int:
            if self._early_discharge is None:
                self._early_discharge = self._early_discharge_func(self.vessel_idx)
            return int(self._early_discharge)

        def __getstate__(self):
            return {"tick": self.tick, "port_idx": self.port_idx, "vessel_idx": self.vessel_idx,
                    "action_scope": self.action_scope, "early_discharge": self.early_discharge}

        def __setstate__(self, state):
            self.tick = state["tick"]
            self.port_idx = state["port_idx"]
            self.vessel_idx = state["vessel_idx"]
            self._action_scope = state["action_scope"]
            self._early_discharge = state["early_discharge"]

        def __repr__(self):
            return "%s {port_idx: %r, vessel_idx: %r, action_scope: %r, early_discharge: %r}" % (
                self.__class__.__name__, self.port_idx, self.vessel_idx, self.action_scope, self.early_discharge)


def encode_action(a) -> tuple:
    """Action object -> the C ABI's (vessel_idx, port_idx, quantity, 0=LOAD | 1=DISCHARGE)."""
    t = a.action_type
    name = getattr(t, "name", str(t)).upper()
    return (int(a.vessel_idx), int(a.port_idx), int(a.quantity), 1 if name.endswith("DISCHARGE") else 0)


_ACTION_CODE = {ActionType.LOAD: 0, ActionType.DISCHARGE: 1}


def action_code(t) -> int:
    """ActionType -> the C ABI's 0 = LOAD | 1 = DISCHARGE (a foreign enum with the same member names is accepted)."""
    c = _ACTION_CODE.get(t)
    if c is None:
        c = 1 if getattr(t, "name", str(t)).upper().endswith("DISCHARGE") else 0
    return c


def make_decision_event(row, snapshot_list) -> "DecisionEvent":
    """Decision row of the C ABI -> DecisionEvent (scope / early discharge already evaluated at the pause)."""
    tick, port, vessel, load, discharge, early = (int(x) for x in row[:6])
    if not HAVE_MARO:
        return DecisionEvent._from_row(tick, port, vessel, snapshot_list, load, discharge, early)
    return DecisionEvent(tick, port, vessel, snapshot_list, lambda p, v: ActionScope(load, discharge), lambda v: early)
