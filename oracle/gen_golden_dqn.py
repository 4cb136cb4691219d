"""TEST INFRASTRUCTURE — golden vectors for the DQN policy pieces, produced by the REAL reference code.

Builds the reference's ``MyQNet`` (examples/cim/rl/algorithms/dqn.py:25-52, over maro/rl/model/fc_block.py) with
dueling heads, randomises every parameter *and* the BatchNorm running statistics, and records in eval mode
(ValueBasedPolicy acts with the net in eval mode): the state_dict, a batch of states, the q-values of
``DiscreteQNet.q_values_for_all_actions`` and the greedy actions.  maro.rl imports zmq/tornado for its distributed training
proxies (absent here); they are stubbed — nothing of them is on this path.

    cp -r /root/reference /tmp/oracle/maro_src        # (built as for oracle/gen_golden.py)
    python3 oracle/gen_golden_dqn.py --maro /tmp/oracle/maro_src --out tests/golden
"""
import argparse
import importlib.util
import os
import sys
from unittest.mock import MagicMock

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--maro", default="/tmp/oracle/maro_src")
    ap.add_argument("--out", default="tests/golden")
    args = ap.parse_args()
    os.environ.setdefault("HOME", "/tmp/oracle/home")
    sys.path.insert(0, args.maro)
    for name in ["zmq", "zmq.asyncio", "zmq.eventloop", "zmq.eventloop.zmqstream", "tornado", "tornado.ioloop"]:
        sys.modules[name] = MagicMock()
    import torch

    spec = importlib.util.spec_from_file_location("ref_dqn", os.path.join(args.maro, "examples/cim/rl/algorithms/dqn.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)

    torch.manual_seed(20240923)
    state_dim, action_num = 45, 21   # (look_back 4 - 1) x (1 + 3 ports) x 3 attrs... a small shape of the same kind
    ref.q_net_conf["hidden_dims"] = [40, 24, 12]
    head = dict(activation=torch.nn.LeakyReLU, softmax=False, batch_norm=True, skip_connection=False, head=True, dropout_p=0.0)
    net = ref.MyQNet(state_dim, action_num, dueling_param=(dict(hidden_dims=[20], output_activation=torch.nn.LeakyReLU, **head),
                                                           dict(hidden_dims=[20], output_activation=None, **head)))
    with torch.no_grad():
        for name, t in net.state_dict().items():
            if name.endswith("running_var"):
                t.copy_(torch.rand_like(t) * 3 + 0.2)
            elif name.endswith("num_batches_tracked"):
                continue
            elif name.endswith("batch_norm.weight"):
                t.copy_(torch.rand_like(t) + 0.5)
            else:
                t.copy_(torch.randn_like(t) * (0.4 if "linear.weight" in name else 1.0))
    net.eval()
    states = torch.randn(64, state_dim) * 5 + 2
    with torch.no_grad():
        q = net.q_values_for_all_actions(states)
    out = {"state_dim": state_dim, "action_num": action_num, "hidden": np.array([40, 24, 12]), "head_hidden": 20,
           "states": states.numpy(), "q": q.numpy(), "greedy": q.argmax(dim=1).numpy()}
    for name, t in net.state_dict().items():
        out["sd:" + name] = t.numpy()
    path = os.path.join(args.out, "dqn_myqnet_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
