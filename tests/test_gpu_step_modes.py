"""The launch forms of mrx_cim_step on an MI355X (mrx_cim_set_step_mode): 1 unsorted, 2 sorted by mrx_k_cim_schedule, 4 split
(plan-specialised kernels), 5 = the fast kernel of 4 + one workgroup per full-path entry (generic and "S5" specialised), and "S" = the plan-specialised LEAN build (return ring + order quantities in registers) in the
sorted launch.  Pure scheduling / pure code generation: outputs and engine state must be identical, and every form must
replay the reference goldens."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _engines(topo, n, modes, **kw):
    import torch

    from maro_amd.cim.engine import CimBatchEngine
    seeds = torch.arange(n, dtype=torch.int64) * 3 + 1
    engs = []
    for m in modes:
        sm = {"S": 2, "S5": 5}.get(m, m)
        e = CimBatchEngine(topo, n, seeds=seeds, specialize=(m in ("S", 4, "S5")), step_mode=sm, **kw)
        assert e.step_mode == sm and e.specialized == (m in ("S", 4, "S5")), (e.step_mode, m)
        engs.append(e)
    return engs


@pytest.mark.parametrize("topo,n", [("global_trade.22p_l0.8", 3000), ("toy.5p_ssddd_l0.5", 777)])
def test_step_modes_are_pure_scheduling(topo, n):
    import torch
    engs = _engines(topo, n, (1, 2, "S", 4, 5, "S5"), durations=100, max_snapshots=5)
    obs = [e.set_observation(["empty", "full", "shortage", "transfer_cost"], ["empty", "remaining_space"]) for e in engs]
    assert [e.step_mode for e in engs] == [1, 2, 2, 4, 5, 5]
    acts = [torch.zeros((n, 1, 4), dtype=torch.int32, device="cuda") for _ in engs]
    nact = [torch.zeros(n, dtype=torch.int32, device="cuda") for _ in engs]
    for e in engs:
        e.step()
    g = torch.Generator(device="cpu").manual_seed(0)
    for k in range(330):
        mask = None
        if k % 5 == 2:   # some envs sit a step out (the same ones in every engine)
            mask = (torch.rand(n, generator=g) < 0.9).to(torch.uint8).cuda()
        if k == 150:     # a third of the batch starts a new episode
            cmd = torch.full((n,), -2, dtype=torch.int64)
            rm = (torch.arange(n) % 3 == 0).to(torch.uint8)
            for e in engs:
                e.reset(cmd, rm)
                e.step(mask=rm)
        for e, a, c in zip(engs, acts, nact):
            e.random_policy(-1, a, c)
            e.step(a, c, mask=mask)
        torch.cuda.synchronize()
        for e, o in zip(engs[1:], obs[1:]):
            assert torch.equal(engs[0].decisions, e.decisions) and torch.equal(engs[0].metrics, e.metrics), (k, e.step_mode)
            assert torch.equal(engs[0].done, e.done), (k, e.step_mode)
            live = engs[0].decisions[:, 7] == 1
            assert torch.equal(obs[0][0][live], o[0][live]) and torch.equal(obs[0][1][live], o[1][live]), (k, e.step_mode)
    for e in engs[1:]:
        assert torch.equal(engs[0].live, e.live) and torch.equal(engs[0].ring, e.ring) and torch.equal(engs[0].ring_fi, e.ring_fi)
        assert torch.equal(engs[0].status, e.status) and torch.equal(engs[0].ticks, e.ticks)
    assert int(engs[0].status.max()) == 0


@pytest.mark.parametrize("mode", [2, "S", 4, "S5"])
@pytest.mark.parametrize("name", ["gt22p_l08_rand0", "toy4p_l03_res7_ring5", "gt22p_l08_res3", "toy6p_l08_rand0", "case_config_folder_kat",
                                  "real_csv_rand0", "gt22p_l08_reset_chain", "syn_repeated_ports_noisy"])
def test_goldens_replay_in_every_launch_form(name, mode):
    from maro_amd.cim.engine import CimBatchEngine
    from tests.backend_adapter import SingleEnvAdapter
    from tests.gpu_backend import GpuBackend
    from tests.test_oracle_golden import replay_case

    def make(topo, kwargs):
        b = GpuBackend.__new__(GpuBackend)
        b.eng = CimBatchEngine(topo, 5, start_tick=kwargs.get("start_tick", 0), durations=kwargs["durations"], snapshot_resolution=kwargs.get("snapshot_resolution", 1),
                               max_snapshots=kwargs.get("max_snapshots"), max_actions=2, specialize=(mode in ("S", 4, "S5")), step_mode={"S": 2, "S5": 5}.get(mode, mode))
        assert b.eng.step_mode == {"S": 2, "S5": 5}.get(mode, mode)
        b.topo, b.layout, b.n_envs, b.max_actions, b.max_tick = b.eng.topo, b.eng.layout, 5, 2, kwargs.get("start_tick", 0) + kwargs["durations"]
        return SingleEnvAdapter(b, env=3)
    replay_case(make, name)


def test_joint_mode_uses_the_order_list_too():
    """Joint decision modes have no fast path: every env is a full-path entry of the order list (sorted launch forms)."""
    import torch
    topo, n = "toy.6p_sssbdd_l0.8", 200
    engs = _engines(topo, n, (1, 2, "S", 4, "S5"), durations=80, decision_mode=1, max_actions=6)
    for e in engs:
        e.step()
    for k in range(60):
        for e in engs:
            e.step()   # Joint mode with no answers: every pending event is finished without an action
        torch.cuda.synchronize()
        for e in engs[1:]:
            assert torch.equal(engs[0].decisions, e.decisions) and torch.equal(engs[0].metrics, e.metrics), (k, e.step_mode)
    for e in engs[1:]:
        assert torch.equal(engs[0].live, e.live) and torch.equal(engs[0].ring, e.ring)
