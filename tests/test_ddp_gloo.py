"""CPU, world_size 2, gloo: the data-parallel gradient exchange (vibravox_amd/ddp.py) that bench.py
uses with RCCL on the GPUs -- bucketed all-reduce with gradients living in the bucket buffers,
launched from post-accumulate hooks, averaged by the optimizer's grad_scale."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vibravox_amd.ddp import BucketedZeroGrad, GradSync, all_reduce_scalars

        torch.manual_seed(0)  # identical weights on both ranks
        net = torch.nn.Sequential(torch.nn.Linear(7, 33), torch.nn.Tanh(), torch.nn.Linear(33, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
        ref = torch.nn.Sequential(torch.nn.Linear(7, 33), torch.nn.Tanh(), torch.nn.Linear(33, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
        ref.load_state_dict(net.state_dict())
        sync = GradSync(net.parameters(), bucket_bytes=600)  # forces several buckets
        assert len(sync.buckets) >= 3
        opt = BucketedZeroGrad(torch.optim.SGD(net.parameters(), lr=0.1), sync)
        ref_opt = torch.optim.SGD(ref.parameters(), lr=0.1)
        g = torch.Generator().manual_seed(100)
        data = torch.randn(2, 4, 7, generator=g)  # same on both ranks; rank r trains on data[r]
        for step in range(3):
            x = data[rank] + step
            net(x).pow(2).sum().backward()
            scale = sync.finish()
            assert scale == 1.0 / world
            for p in net.parameters():
                p.grad.mul_(scale)
            opt.step()
            opt.zero_grad()
            # single-process reference: mean of the per-rank gradients
            ref_opt.zero_grad()
            sum(ref(data[r] + step).pow(2).sum() for r in range(world)).backward()
            for p in ref.parameters():
                p.grad.div_(world)
            ref_opt.step()
            for p, r in zip(net.parameters(), ref.parameters()):
                assert torch.allclose(p, r, atol=1e-6), (step, (p - r).abs().max())
            assert all(p.grad is v for b in sync.buckets for p, v in zip(b.params, b.views))  # grads stay bucket views
        assert sync.launched >= 3 * len(sync.buckets)  # every bucket went out from its hook, every step
        # the sink form the engine step uses: gradients computed outside autograd are written straight into the bucket views
        # (grad_buffer) and reported with mark_ready -- same exchange, same result
        launched = sync.launched
        x = data[rank] + 7
        grads = torch.autograd.grad(net(x).pow(2).sum(), list(net.parameters()))
        plist = list(net.parameters())
        for p, gr in zip(plist, grads):
            buf = sync.grad_buffer(p)
            assert buf is not None and buf.shape == p.shape and buf.is_contiguous()
            buf.copy_(gr)
        sync.mark_ready(reversed(plist))
        scale = sync.finish()
        assert sync.launched == launched + len(sync.buckets)
        want = torch.autograd.grad(sum(ref(data[r] + 7).pow(2).sum() for r in range(world)), list(ref.parameters()))
        for p, w in zip(plist, want):
            assert p.grad is sync.grad_buffer(p)
            assert torch.allclose(p.grad * scale, w / world, atol=1e-6)
        opt.zero_grad()
        vals = all_reduce_scalars([torch.tensor(float(rank)), torch.tensor(10.0 + rank)])
        assert abs(float(vals[0]) - 0.5) < 1e-6 and abs(float(vals[1]) - 10.5) < 1e-6
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, f"{type(e).__name__}: {e}"))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_bucketed_gradient_allreduce_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert sorted(results) == [(0, "ok"), (1, "ok")], results
