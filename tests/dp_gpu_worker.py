"""One rank of tests/test_gpu_dp_two_ranks.py: the REAL Trainer.step (HIP kernels, autograd hook, communication stream,
flat buckets) on this rank's shard of a fixed global batch, world_size ranks over gloo on ONE GPU (RCCL refuses two ranks
on one device; gloo all-reduces the same CUDA bucket tensors).  Also imported by the test for the single-process run.

usage: python tests/dp_gpu_worker.py <rank> <world> <port> <case> <out.npz>"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

L, A, R, C, H, W, B, HORIZON = 16, 5, 2, 3, 16, 16, 4, 4


def batch(gan):
    from scm_gan_amd.datasource import SyntheticDatasource
    ds = SyntheticDatasource(seed=3, height=H, width=W, channels=C, num_actions=A, num_rewards=R, done_every=3)
    states, rewards, dones, actions = ds.get_trajectories(B, HORIZON)
    rng = np.random.default_rng(17)
    noises = [rng.random((B, L, H, W), dtype=np.float32) for _ in range(16)]
    dnoise = (rng.random((B, 3, 64, 64)) < 0.5).astype(np.float32) if gan else None
    return states, rewards, dones.astype(np.float32), actions, noises, dnoise


def run_steps(case, shard, steps=2):
    """Trainer.step x steps on samples `shard`; returns (reduced flat gradients of the last step, weights, hook count, enabled)."""
    from scm_gan_amd import main as M, models
    dev = torch.device("cuda", 0)
    gan = case == "gan"
    states, rewards, dones, actions, noises, dnoise = batch(gan)
    torch.manual_seed(0)                                   # identical replicas
    enc, dec, rp = models.Encoder(L, C), models.Decoder(L, C), models.RewardPredictor(L, R)
    disc, tr = models.Discriminator(), models.Transition(L, A)
    flags = {"default": [], "gan": ["--gan"], "no_overlap": ["--no-overlap"]}[case]
    opts = M.parser.parse_args(["--train-iters", "10", "--horizon-min", str(HORIZON), "--horizon-max", str(HORIZON)] + flags)
    trainer = M.Trainer(L, A, R, enc, dec, rp, disc, tr, opts)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a[shard])).to(dev)
    oh = M.onehot(actions[shard], A)
    for it in range(steps):
        tr.set_noise([torch.from_numpy(n[shard]).to(dev) for n in noises])
        trainer.step(t(states), t(rewards), t(dones), oh, 0.5, it + 1, actions[shard],
                     disc_noise=None if dnoise is None else torch.from_numpy(dnoise[shard]).to(dev))
    torch.cuda.synchronize()
    # scm_adam_clip_multi stores the scaled (1 / world) and clamped gradient back, as clip_grad_value_ does: after the
    # step the buckets hold clamp(sum over ranks / world), to be compared with the single process's clamp(gradient)
    grads = torch.cat([f.flat.clone() for f in trainer.flats]).cpu()
    weights = torch.cat([p.detach().flatten() for m in (enc, dec, tr, rp, disc) for p in m.parameters()]).cpu()
    return grads, weights, trainer.sync.hook_fired, trainer.sync.enabled


if __name__ == "__main__":
    rank, world, port, case, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    half = B // world
    g, w, fired, enabled = run_steps(case, slice(rank * half, (rank + 1) * half))
    np.savez(out, g=g.numpy(), w=w.numpy(), fired=fired, enabled=int(enabled))
    dist.destroy_process_group()
