"""The CUDA branch of train_step.GradientBucket (side stream, events, pinned participation flags) with TWO ranks: both
processes share the one GPU of the test box and talk over gloo (RCCL refuses two ranks on one device; gloo stages CUDA
tensors through the host), which exercises exactly the code a multi-GPU run takes except the transport."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import datetime
    import torch.distributed as dist
    from regnet_for_3d_grasping_amd import train_step
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.ReLU(), torch.nn.Linear(64, 8)).cuda()
    unused = torch.nn.Linear(4, 4).cuda()                     # never receives a gradient on any rank
    only0 = torch.nn.Linear(8, 1).cuda()                      # receives one on rank 0 only
    train_step.broadcast_module_state(net, unused, only0)
    bucket = train_step.GradientBucket([net, unused, only0])
    opt = torch.optim.Adam(list(net.parameters()) + list(unused.parameters()) + list(only0.parameters()), lr=1e-2)
    calls = []
    orig = dist.all_reduce
    dist.all_reduce = lambda t, *a, **k: (calls.append(int(t.numel())), orig(t, *a, **k))[1]
    results = []
    for step in range(3):
        torch.manual_seed(100 * step + rank)
        x = torch.randn(16, 32, device="cuda")
        bucket.prepare()
        y = net(x)
        loss = y.pow(2).mean()
        if rank == 0:
            loss = loss + only0(y).mean()
        loss.backward()
        local = {k: p.grad.clone() for k, p in net.named_parameters()}
        bucket.reduce_gradients()
        results.append({"local": {k: v.cpu() for k, v in local.items()},
                        "reduced": {k: p.grad.clone().cpu() for k, p in net.named_parameters()},
                        "only0_has_grad": only0.weight.grad is not None, "unused_has_grad": unused.weight.grad is not None,
                        "only0_grad": None if only0.weight.grad is None else only0.weight.grad.clone().cpu()})
        opt.step()
    torch.cuda.synchronize()
    torch.save({"results": results, "calls": calls, "n": [len(bucket.params), bucket.n_grad], "ms": bucket.last_allreduce_ms(),
                "params": {k: p.detach().cpu() for k, p in net.named_parameters()},
                "only0": only0.weight.detach().cpu()}, os.path.join(out_dir, "r%d.pt" % rank))
    dist.destroy_process_group()


def test_gradient_bucket_cuda_branch_two_ranks_one_gpu(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, "r%d.pt" % r)) for r in range(2))
    assert r0["calls"] == r1["calls"] == r0["n"] * 3                   # per step: the presence flags, then ONE gradient collective; fixed lengths
    assert r0["ms"] is not None and r0["ms"] >= 0.0                     # event-timed
    for a, b in zip(r0["results"], r1["results"]):
        for k in a["local"]:
            want = a["local"][k] + b["local"][k]
            torch.testing.assert_close(a["reduced"][k], want, rtol=0, atol=1e-6)
            assert torch.equal(a["reduced"][k], b["reduced"][k])
        assert a["only0_has_grad"] and b["only0_has_grad"]              # rank 1 receives rank 0's gradient
        assert torch.equal(a["only0_grad"], b["only0_grad"])
        assert not a["unused_has_grad"] and not b["unused_has_grad"]    # nobody touched it: stays grad-less
    for k in r0["params"]:
        assert torch.equal(r0["params"][k], r1["params"][k])            # replicas stay identical
    assert torch.equal(r0["only0"], r1["only0"])
