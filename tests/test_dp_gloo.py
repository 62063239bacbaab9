"""CPU, world_size 2 over gloo: the data-parallel host logic of Trainer -- flat
gradient buckets (one all-reduce per network, grads are views of the bucket),
SUM all-reduce followed by the 1/world scale == the single-process gradient of
the full batch, and identical replicas after the update."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _net(seed):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.LeakyReLU(), torch.nn.Conv2d(8, 2, 3, padding=1))


def _loss(net, x, y):
    return ((net(x) - y) ** 2).mean(dim=(1, 2, 3)).mean()      # mean over the (local) batch, like main.py:183,195


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scm_gan_amd.main import _FlatGrads, _dist
    assert _dist() == (rank, world)
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.standard_normal((8, 3, 6, 6)).astype(np.float32))
    y = torch.from_numpy(rng.standard_normal((8, 2, 6, 6)).astype(np.float32))
    net = _net(1)                                      # identical replicas
    flat = _FlatGrads(net)
    flat.zero()
    shard = slice(rank * 4, rank * 4 + 4)              # equal shards of the global batch
    _loss(net, x[shard], y[shard]).backward()
    for p in net.parameters():                         # autograd accumulated IN the bucket views
        assert p.grad.data_ptr() >= flat.flat.data_ptr()
    flat.all_reduce()
    g = flat.flat / world                              # AdamClip.grad_scale = 1/world, applied before the clip
    with torch.no_grad():
        for p in net.parameters():
            p -= 0.1 * (p.grad / world).clamp(-0.1, 0.1)
    out[rank] = (g.clone(), torch.cat([p.detach().flatten() for p in net.parameters()]))
    dist.destroy_process_group()


def test_flat_bucket_allreduce_equals_full_batch():
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    # single process, full batch
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.standard_normal((8, 3, 6, 6)).astype(np.float32))
    y = torch.from_numpy(rng.standard_normal((8, 2, 6, 6)).astype(np.float32))
    net = _net(1)
    _loss(net, x, y).backward()
    ref = torch.cat([p.grad.flatten() for p in net.parameters()])
    for r in range(world):
        assert torch.allclose(out[r][0], ref, atol=1e-6), (out[r][0] - ref).abs().max()
    assert torch.equal(out[0][1], out[1][1])           # replicas stay bit-identical


def test_bench_refuses_wrong_launch(monkeypatch):
    """bench.py --gpus N must be launched one rank per GPU."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                       capture_output=True, text=True, env={**os.environ, "WORLD_SIZE": "1"})
    assert r.returncode != 0 and "torch.distributed.run" in (r.stderr + r.stdout)
