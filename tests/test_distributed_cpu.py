"""world_size-2 gloo tests (CPU) of text2speech_amd/distributed.py: the three reference entry points."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    import torch.distributed as dist
    from text2speech_amd import distributed as D
    D.init_distributed(rank, world, None, "gloo", "tcp://127.0.0.1:%d" % port)
    try:
        # reduce_tensor: mean over ranks (reference distributed.py:37-41)
        t = torch.tensor([float(rank + 1)])
        r = D.reduce_tensor(t, world)
        assert abs(float(r) - 1.5) < 1e-6 and float(t) == rank + 1
        # apply_gradient_allreduce: same object back, weights broadcast from rank 0, grads averaged
        torch.manual_seed(100 + rank)
        m = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 4))
        m2 = D.apply_gradient_allreduce(m)
        assert m2 is m
        w = [p.detach().clone() for p in m.parameters()]
        gathered = [torch.zeros_like(w[0]) for _ in range(world)]
        dist.all_gather(gathered, w[0])
        assert torch.equal(gathered[0], gathered[1])
        torch.manual_seed(7 + rank)
        x = torch.randn(5, 8)
        m(x).pow(2).sum().backward()
        g = m[0].weight.grad.clone()
        # expected: the mean of the per-rank local gradients
        ref = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 4))
        ref.load_state_dict(m.state_dict())
        want = torch.zeros_like(g)
        for r_ in range(world):
            torch.manual_seed(7 + r_)
            xr = torch.randn(5, 8)
            ref.zero_grad()
            ref(xr).pow(2).sum().backward()
            want += ref[0].weight.grad / world
        assert torch.allclose(g, want, atol=1e-6), float((g - want).abs().max())
        # GradSync buckets: async all-reduce + finish gives the mean (gloo has no AVG: SUM then scale)
        gs = D.GradSync()
        flat = torch.full((1000,), float(rank))
        gs.reduce_async(flat)
        gs.finish()
        assert torch.allclose(flat, torch.full((1000,), 0.5))
        # the shape the hand-written WaveGlow backward uses (glow_autograd._Bucket): several flat buckets shipped one after
        # the other while later work is still being produced, gradients are VIEWS into the buckets, one finish() at the end
        gs = D.GradSync()
        buckets, views = [], []
        for k in range(5):
            flat = torch.empty(64 * (k + 1) + 8)
            a = flat[:64 * (k + 1)].view(k + 1, 64)
            b = flat[64 * (k + 1):]
            a.fill_(float(rank + k))
            b.fill_(float(10 * rank))
            gs.reduce_async(flat)
            buckets.append(flat)
            views.append((a, b))
        assert gs.n_buckets == 5 and len(gs.pending) == 5
        gs.finish()
        assert not gs.pending
        for k, (a, b) in enumerate(views):
            assert torch.allclose(a, torch.full_like(a, k + 0.5)) and torch.allclose(b, torch.full_like(b, 5.0))
        # the diagnostics bench.py reports for an N > 1 run (per-bucket bytes; timing needs device events, absent on the host)
        st = gs.stats()
        assert st["world"] == world and st["backend"] == "gloo" and [b["bytes"] for b in st["buckets"]] == [4 * (64 * (k + 1) + 8) for k in range(5)]
        out.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        out.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
