"""CPU, world_size 2 over gloo: the data-parallel host logic.

  * dp.FlatGrads: grads are views of one bucket; SUM all-reduce + 1/world == the full-batch gradient;
  * the REAL scm_gan_amd.main.Trainer.step on two ranks with B/2 samples each (networks and HIP leaf
    operators replaced by the oracle's arithmetic, torch.cuda streams stubbed: tests/cpu_backend.py)
    == one process with the full batch: reduced gradients, post-Adam weights, identical replicas; the
    early buckets are issued from the autograd hook on the first latent (GradSync), also with the
    extra encoder / transition calls of --latent-overshooting and with the --gan discriminator pass;
  * main.train() across an evaluation boundary keeps SpectralNorm's u / v identical on all ranks;
  * bench.py --gpus N starts its own ranks and fails with a clear message when the GPUs are missing.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _paths():
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    _paths()
    torch.set_num_threads(2)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


# ------------------------------------------------------------------ flat bucket arithmetic
def _net(seed):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.LeakyReLU(), torch.nn.Conv2d(8, 2, 3, padding=1))


def _loss(net, x, y):
    return ((net(x) - y) ** 2).mean(dim=(1, 2, 3)).mean()      # mean over the (local) batch, like main.py:183,195


def _bucket_worker(rank, world, port, out):
    _init(rank, world, port)
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
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_bucket_worker, args=(world, _port(), out), nprocs=world, join=True)
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.standard_normal((8, 3, 6, 6)).astype(np.float32))
    y = torch.from_numpy(rng.standard_normal((8, 2, 6, 6)).astype(np.float32))
    net = _net(1)
    _loss(net, x, y).backward()
    ref = torch.cat([p.grad.flatten() for p in net.parameters()])
    for r in range(world):
        assert torch.allclose(out[r][0], ref, atol=1e-6), (out[r][0] - ref).abs().max()
    assert torch.equal(out[0][1], out[1][1])           # replicas stay bit-identical


# ------------------------------------------------------- the real Trainer.step on two ranks
B, H, W, HORIZON, A, R, C, L = 4, 12, 12, 5, 5, 2, 3, 16


def _batch(gan):
    """The global batch + per-call latent noise (+ discriminator noise maps), from fixed seeds."""
    from scm_gan_amd.datasource import SyntheticDatasource
    ds = SyntheticDatasource(seed=3, height=H, width=W, channels=C, num_actions=A, num_rewards=R, done_every=3)
    states, rewards, dones, actions = ds.get_trajectories(B, HORIZON)
    rng = np.random.default_rng(17)
    noises = [rng.random((B, L, H, W), dtype=np.float32) for _ in range(16)]
    dnoise = (rng.random((B, 3, 64, 64)) < 0.5).astype(np.float32) if gan else None
    return states, rewards, dones.astype(np.float32), actions, noises, dnoise


def _flags(case):
    cf = ["--enable-disentanglement-loss", "--enable-action-control-loss"]
    return {"default": [], "lo": ["--latent-overshooting"], "gan": ["--gan"], "no_overlap": ["--no-overlap"],
            "cf": cf, "cf_no_overlap": cf + ["--no-overlap"]}[case]


def _run_step(case, shard, steps=2, batched_heads=False):
    """Trainer.step x `steps` on the samples `shard` of the global batch; returns (reduced flat grads of
    the last step, weights after it, hook count).  batched_heads: the reward head and the decoder run once over all
    unrolled steps (main._heads_time_batched) instead of once per step."""
    import cpu_backend as CB
    M = CB.install()
    os.environ["SCM_BATCHED_HEADS"] = "1" if batched_heads else "0"
    gan = case == "gan"
    states, rewards, dones, actions, noises, dnoise = _batch(gan)
    enc, dec, rp, disc, tr = CB.build_nets(0, A, R, C, L)
    opts = M.parser.parse_args(["--train-iters", "10", "--horizon-min", str(HORIZON), "--horizon-max", str(HORIZON)]
                               + _flags(case))
    trainer = M.Trainer(L, A, R, enc, dec, rp, disc, tr, opts)
    np.random.seed(1234)            # (the counterfactual regularisers draw factor pairs / action shuffles from numpy's global stream)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a[shard]))
    oh = torch.eye(A)[torch.from_numpy(actions[shard])]
    for it in range(steps):
        tr.set_noise([torch.from_numpy(n[shard]) for n in noises])
        trainer.step(t(states), t(rewards), t(dones), oh, 0.5, it + 1, actions[shard],
                     disc_noise=None if dnoise is None else torch.from_numpy(dnoise[shard]))
    os.environ.pop("SCM_BATCHED_HEADS", None)
    grads = torch.cat([f.flat.clone() for f in trainer.flats]) / trainer.world
    weights = torch.cat([p.detach().flatten() for m in (enc, dec, tr, rp, disc) for p in m.parameters()])
    return grads, weights, trainer.sync.hook_fired, trainer


def _trainer_worker(rank, world, port, case, out):
    _init(rank, world, port)
    half = B // world
    # (case "batched": the two ranks run the time-batched heads, the one-process reference below the per-step launches)
    grads, weights, fired, trainer = _run_step("default" if case == "batched" else case,
                                               slice(rank * half, (rank + 1) * half), batched_heads=case == "batched")
    out[rank] = (grads, weights, fired, trainer.sync.enabled)
    dist.destroy_process_group()


@pytest.mark.parametrize("case", ["default", "lo", "gan", "no_overlap", "batched"])
def test_trainer_step_two_ranks_equal_full_batch(case):
    _paths()
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_trainer_worker, args=(world, _port(), case, out), nprocs=world, join=True)
    ref_g, ref_w, fired, trainer = _run_step("default" if case == "batched" else case, slice(0, B))   # one process, the full batch
    assert not trainer.sync.enabled and fired == 0                     # no collective without a process group
    for r in range(world):
        g, w, fired, enabled = out[r]
        assert enabled
        # overlap: the hook on the first latent's gradient issued the early buckets in both iterations
        assert fired == (0 if case == "no_overlap" else 2), fired
        assert torch.allclose(g, ref_g, atol=2e-6, rtol=1e-4), (case, float((g - ref_g).abs().max()))
        # Adam moves a weight by <= lr = 1e-4 per step whatever |g|: a ~0 gradient may flip sign
        err = (w - ref_w).abs()
        assert float(err.max()) <= 4.1e-4 and float((err < 2e-6).float().mean()) > 0.995, (case, float(err.max()))
    assert torch.equal(out[0][1], out[1][1])                           # replicas stay bit-identical


def _cf_worker(rank, world, port, out):
    _init(rank, world, port)
    half = B // world
    shard = slice(rank * half, (rank + 1) * half)
    res = {}
    for case in ("cf", "cf_no_overlap"):
        # five iterations: the counterfactual terms enter on train_iter % CF_REGULARIZATION_RATE == 0 (main.py:242,268)
        grads, weights, fired, trainer = _run_step(case, shard, steps=5)
        assert trainer.a.enable_disentanglement_loss and trainer.a.enable_action_control_loss
        res[case] = (grads, weights, fired)
    out[rank] = res
    dist.destroy_process_group()


def test_counterfactual_losses_overlapped_buckets_equal_no_overlap():
    """GradSync's hook relies on the autograd engine finishing every node created after the first
    encoder call before the first latent's gradient arrives.  The counterfactual regularisers (main.py:242-283) add
    consumers of that latent -- z_orig = z.clone(), mutated in place, rolled forward twice more through the
    transition -- AFTER the unrolled loop: with them on, the buckets reduced early from the hook must equal the
    buckets reduced after backward() (--no-overlap), on both ranks, in the iteration where the terms are active."""
    _paths()
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_cf_worker, args=(world, _port(), out), nprocs=world, join=True)
    for r in range(world):
        g_o, w_o, fired_o = out[r]["cf"]
        g_n, w_n, fired_n = out[r]["cf_no_overlap"]
        assert fired_o == 5 and fired_n == 0
        assert float(g_o.abs().max()) > 0
        assert torch.equal(g_o, g_n), float((g_o - g_n).abs().max())
        assert torch.equal(w_o, w_n)
    assert torch.equal(out[0]["cf"][1], out[1]["cf"][1])               # replicas stay bit-identical


# ----------------------------------------------------- u / v across an evaluation boundary
def _train_worker(rank, world, port, tmp, out):
    _init(rank, world, port)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    os.chdir(tmp)
    import cpu_backend as CB
    from scm_gan_amd import dp
    from scm_gan_amd.datasource import SyntheticDatasource
    M = CB.install()
    enc, dec, rp, disc, tr = CB.build_nets(0, A, R, C, L)
    rng = np.random.default_rng(100 + rank)
    tr.set_noise([torch.from_numpy(rng.random((2, L, H, W), dtype=np.float32)) for _ in range(64)])
    M.args = M.parser.parse_args(["--batch-size", "4", "--train-iters", "4", "--horizon-min", "4", "--horizon-max", "4",
                                  "--eval-every", "2", "--log-every", "1", "--check-replicas"])
    ds = SyntheticDatasource(seed=5 + dp.SEED_STRIDE * rank, height=H, width=W, channels=C, num_actions=A, num_rewards=R)
    _mse = M.measure_prediction_mse
    M.measure_prediction_mse = lambda *a, **k: _mse(*a, **dict(k, timesteps=5, batch_size=3))   # keep the eval rollout short
    M.train(L, ds, A, R, enc, dec, rp, disc, tr)
    uv = torch.cat([p.detach().flatten() for m in (enc, tr) for k, p in m.p.items() if k.endswith("weight_u") or k.endswith("weight_v")])
    w = torch.cat([p.detach().flatten() for m in (enc, dec, tr, rp) for p in m.parameters()])
    out[rank] = (uv, w, sorted(os.listdir(tmp)))
    dist.destroy_process_group()


def test_train_keeps_replicas_identical_across_eval(tmp_path):
    """SpectralNorm iterates on every forward, eval included; with rank-0-only evaluation the
    ranks' u / v (and so their effective weights) drifted apart.  Now every rank evaluates, rank 0 writes."""
    _paths()
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_train_worker, args=(world, _port(), str(tmp_path), out), nprocs=world, join=True)
    assert torch.equal(out[0][0], out[1][0])            # u / v
    assert torch.equal(out[0][1], out[1][1])            # every parameter
    files = out[0][2]
    assert "model-encoder.pth" in files and "mse_pixel_iter_000002.json" in files
    assert sum(f.startswith("mse_pixel_iter_000002") for f in files) == 1       # written once (rank 0)


# ------------------------------------------------------------------------------- launcher
def test_bench_self_launch_reports_missing_gpus():
    """`python bench.py --gpus 2` (no launcher) spawns its own ranks; on a box with fewer GPUs it must say so
    -- a clear 'needs 2 GPUs' message, not a launcher complaint."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                       capture_output=True, text=True, env=env)
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has the GPUs")
    assert r.returncode != 0 and "needs 2 GPUs on this node" in (r.stderr + r.stdout), r.stderr[-500:]


def test_init_from_env_rejects_half_set_environment(monkeypatch):
    from scm_gan_amd import dp
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.delenv("MASTER_PORT", raising=False)
    with pytest.raises(SystemExit):
        dp.init_from_env()
