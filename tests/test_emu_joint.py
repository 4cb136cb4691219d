"""DecisionMode.Joint / JointWithSequentialAction (core.py:354-366): the device source on the CPU wave emulator
against vectors from the real reference."""
import numpy as np
import pytest

from tests.backend_adapter import SingleEnvAdapter
from tests.emu.emu import EmuBackend
from tests.golden_util import joint_golden_cases, replay_joint_case


class JointAdapter(SingleEnvAdapter):
    """step_joint(actions, n_answered) over a batch backend created with decision_mode 1 / 2."""

    def step_joint(self, actions, n_answered):
        b = self.b
        acts = np.full((b.n_envs, b.max_actions, 4), -1, np.int32)
        na = np.zeros(b.n_envs, np.int32)
        if actions:
            assert len(actions) <= b.max_actions
            for i, a in enumerate(actions):
                acts[:, i] = a
            na[:] = len(actions)
        dec, met, done = b.step(acts, na, n_answered=np.full(b.n_envs, n_answered, np.int32))
        assert (dec == dec[self.e]).all() and (met == met[self.e]).all()
        self._paused = not bool(done[self.e])
        return met[self.e], dec[self.e], bool(done[self.e])


def make_emu(order_table=0, reverse=False):
    def make(topo, kwargs, mode):
        b = EmuBackend(topo, n_envs=2, start_tick=kwargs.get("start_tick", 0), durations=kwargs["durations"], snapshot_resolution=kwargs.get("snapshot_resolution", 1),
                       max_snapshots=kwargs.get("max_snapshots"), max_actions=topo.n_vessels, decision_mode=mode,
                       order_table=order_table, reverse=reverse)
        return JointAdapter(b, env=1)
    return make


@pytest.mark.parametrize("name", joint_golden_cases())
def test_joint_modes_on_emulated_kernels(name):
    replay_joint_case(make_emu(), name)


def test_joint_mode_online_orders_and_reverse_lanes():
    replay_joint_case(make_emu(order_table=-1, reverse=True), "jointseq_toy5p_l05_some")
