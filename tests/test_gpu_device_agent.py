"""mrx_cim_set_device_agent on the HIP engine: the random legal agent answered inside the step kernel gives, step by step, the
decisions / metrics / done flags / actions of the separate mrx_cim_random_policy launch — generic and plan-specialised kernels,
unsorted and sorted launch, with the fused observation — and (tests/test_emu_device_agent.py) the oracle's trajectories."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
P_ATTRS = ["empty", "full", "on_shipper", "on_consignee", "booking", "shortage", "fulfillment"]
V_ATTRS = ["empty", "full", "remaining_space"]


@pytest.mark.parametrize("specialize,step_mode,obs", [(False, 1, False), (False, 2, True), (True, 1, True), (True, 2, False), (True, 4, True)])
def test_device_agent_equals_the_policy_launch(specialize, step_mode, obs):
    from maro_amd.cim.engine import CimBatchEngine
    n, dur = 700, 160
    seeds = torch.arange(n, dtype=torch.int64) * 3 + 17
    engs = [CimBatchEngine("global_trade.22p_l0.8", n, durations=dur, max_actions=1, max_snapshots=4, seeds=seeds, specialize=specialize, step_mode=step_mode)
            for _ in range(2)]
    obs_t = [e.set_observation(P_ATTRS, V_ATTRS) if obs else None for e in engs]
    acts = [torch.zeros((n, 1, 4), dtype=torch.int32, device="cuda") for _ in engs]
    nact = [torch.zeros(n, dtype=torch.int32, device="cuda") for _ in engs]
    counts, counter = torch.zeros(n, dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.int64, device="cuda")
    engs[1].set_device_agent(acts[1], nact[1], counts, next_key=1)
    outs = [e.step() for e in engs]
    k = 0
    while not bool(engs[0].done.all()):
        k += 1
        engs[0].random_policy(k, acts[0], nact[0], counter)
        torch.cuda.synchronize()
        assert torch.equal(nact[0], nact[1]), k
        live = nact[0] == 1
        assert torch.equal(acts[0][live], acts[1][live]), k
        for x, y in zip(*outs):
            assert torch.equal(x, y), k
        if obs:
            assert torch.equal(obs_t[0][0][live], obs_t[1][0][live]) and torch.equal(obs_t[0][1][live], obs_t[1][1][live]), k
        outs = [e.step(a, m) for e, a, m in zip(engs, acts, nact)]
    assert k > 150 and int(counts.sum()) == int(counter.item())
    for name in ("live", "ring", "ring_fi", "ticks", "status"):
        assert torch.equal(getattr(engs[0], name), getattr(engs[1], name)), name
    # a second episode: the key restarts with set_device_agent
    for e in engs:
        e.reset(seeds + 1)
    engs[1].set_device_agent(acts[1], nact[1], None, next_key=1)
    outs = [e.step() for e in engs]
    for k in range(1, 40):
        engs[0].random_policy(k, acts[0], nact[0])
        for x, y in zip(*outs):
            assert torch.equal(x, y), k
        outs = [e.step(a, m) for e, a, m in zip(engs, acts, nact)]
    engs[1].set_device_agent()
    acts[1].fill_(-5)
    engs[1].step(acts[0], nact[0])
    assert bool((acts[1] == -5).all())
