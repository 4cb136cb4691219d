"""Test harness transport for maro_amd/cim/rollout.py's exchanges: host staging.

The product functions (`gather_to_learner`, `gather_experiences_to_learner`, `broadcast_policy`) post device tensors to
``torch.distributed`` exactly as they are (RCCL over xGMI).  A 1-GPU box cannot run two RCCL ranks on one device, so the
N>1 tests there run under gloo, whose send / recv take host tensors: this transport stages through host memory on both
sides.  It lives under tests/ — nothing in maro_amd/ imports it; bench.py picks it up only under its MRX_BENCH_BACKEND=gloo
test hook."""
import torch

from maro_amd.cim.rollout import Transport


class HostStaging(Transport):
    def outbound(self, t: torch.Tensor) -> torch.Tensor:
        return t.cpu() if t.is_cuda else t

    def landing(self, shape, like: torch.Tensor) -> torch.Tensor:
        return torch.empty(shape, dtype=like.dtype, device="cpu")

    def inbound(self, t: torch.Tensor, device: torch.device) -> torch.Tensor:
        return t.to(device)
