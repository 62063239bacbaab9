"""GPU (-m gpu): the training iteration replayed as one HIP graph (scm_gan_amd/graph_step.py) against the
eager loop -- same kernels, same values, so the weights must be BITWISE equal after every iteration; at the
launch-bound configs[0] size the replay must also be faster."""
import time

import numpy as np
import pytest
import torch

import weights as WS

pytestmark = pytest.mark.gpu


def up(x):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()


def build(seed):
    from scm_gan_amd import models as M
    enc, dec, rp, disc, tr = M.Encoder(16, 3), M.Decoder(16, 3), M.RewardPredictor(16, 2), M.Discriminator(), M.Transition(16, 5)
    enc.load_state_dict(WS.to_torch(WS.encoder_state(seed)))
    tr.load_state_dict(WS.to_torch(WS.transition_state(seed + 1)))
    dec.load_state_dict(WS.to_torch(WS.decoder_state(seed + 2)))
    rp.load_state_dict(WS.to_torch(WS.reward_state(seed + 3)))
    return enc, dec, rp, disc, tr


def weights_of(nets):
    return torch.cat([p.detach().flatten() for m in nets for p in m.parameters()])


def run(graphed, batches, thetas, B, horizon, size, eval_after=None):
    from scm_gan_amd import main as T
    from scm_gan_amd.graph_step import GraphedTrainer
    nets = build(21)
    enc, dec, rp, disc, tr = nets
    tr.seed_noise(77)
    opts = T.parser.parse_args(["--batch-size", str(B), "--train-iters", "100", "--horizon-min", str(horizon),
                                "--horizon-max", str(horizon)])
    trainer = T.Trainer(16, 5, 2, enc, dec, rp, disc, tr, opts)
    stepper = GraphedTrainer(trainer, warmup=2) if graphed else trainer
    snaps, losses = [], []
    for i, ((s, r, d, a), th) in enumerate(zip(batches, thetas)):
        loss = stepper.step(up(s), up(r), up(d.astype(np.float32)), T.onehot(a, 5), th, i + 1, a)
        snaps.append(weights_of((enc, dec, rp, tr)).clone())
        losses.append(loss.clone())
        if eval_after is not None and i == eval_after:
            # what evaluate() does between two training iterations: a no-grad forward through every network
            # (it fills the weight-pack caches for the CURRENT weights and advances SpectralNorm's u / v)
            with torch.no_grad():
                x = up(s)
                z = enc(x[:, :3])
                z = tr(z, T.onehot(a, 5)[:, 2])
                dec(z)
                rp(z)
    torch.cuda.synchronize()
    return snaps, losses, trainer, stepper


def test_graphed_iterations_are_bitwise_the_eager_ones():
    from scm_gan_amd.datasource import SyntheticDatasource
    B, horizon, size, iters = 4, 5, 16, 7
    ds = SyntheticDatasource(seed=4, height=size, width=size, done_every=3)
    batches = [ds.get_trajectories(B, horizon) for _ in range(iters)]
    thetas = [(i + 1) / 10.0 for i in range(iters)]                 # theta changes every iteration (main.py:143)
    eager, el, te, _ = run(False, batches, thetas, B, horizon, size)
    graph, gl, tg, stepper = run(True, batches, thetas, B, horizon, size)
    assert stepper.replays == iters - 2                             # two eager warm-up iterations, then replays
    for i in range(iters):
        assert torch.equal(eager[i], graph[i]), (i, float((eager[i] - graph[i]).abs().max()))
        assert torch.equal(el[i], gl[i]), i
    # logged losses went through the static buffers
    for name in ("Reconstruction t=1", "Rd Loss t=3"):
        assert te.ts.values(name) == tg.ts.values(name)
    # eager code after the replays sees the updated weights (packs rebuilt): one more eager step on both
    s, r, d, a = ds.get_trajectories(B, horizon)
    from scm_gan_amd import main as T
    for t in (te, tg):
        t.step(up(s), up(r), up(d.astype(np.float32)), T.onehot(a, 5), 0.9, iters + 1, a)
    assert torch.equal(weights_of([te.encoder, te.decoder, te.reward_predictor, te.transition]),
                       weights_of([tg.encoder, tg.decoder, tg.reward_predictor, tg.transition]))


def test_capture_right_after_an_eager_forward_rebuilds_the_packs():
    """an eager no-grad forward (an evaluate()) between the last warm-up iteration and the capture leaves
    weight packs of the current weights in the caches; the capture must not reuse them (no pack kernel would be
    recorded and every replay would run on capture-time weights).  Bitwise equality with the eager loop over 4 replays."""
    from scm_gan_amd.datasource import SyntheticDatasource
    B, horizon, size, iters = 4, 5, 16, 6
    ds = SyntheticDatasource(seed=9, height=size, width=size, done_every=3)
    batches = [ds.get_trajectories(B, horizon) for _ in range(iters)]
    thetas = [0.5] * iters
    eager, el, _, _ = run(False, batches, thetas, B, horizon, size, eval_after=1)
    graph, gl, _, stepper = run(True, batches, thetas, B, horizon, size, eval_after=1)
    assert stepper.replays == iters - 2
    for i in range(iters):
        assert torch.equal(eager[i], graph[i]), (i, float((eager[i] - graph[i]).abs().max()))
        assert torch.equal(el[i], gl[i]), i


def test_graph_replay_at_configs0_size():
    """BASELINE configs[0] shape: 64x64, batch 8, horizon 5."""
    from scm_gan_amd import main as T
    from scm_gan_amd.datasource import SyntheticDatasource
    from scm_gan_amd.graph_step import GraphedTrainer
    B, horizon = 8, 5
    ds = SyntheticDatasource(seed=1)
    s, r, d, a = ds.get_trajectories(B, horizon)
    batch = (up(s), up(r), up(d.astype(np.float32)), T.onehot(a, 5))
    times = {}
    for graphed in (False, True):
        nets = build(5)
        nets[4].seed_noise(3)
        opts = T.parser.parse_args(["--batch-size", str(B), "--train-iters", "1000"])
        trainer = T.Trainer(16, 5, 2, nets[0], nets[1], nets[2], nets[3], nets[4], opts)
        stepper = GraphedTrainer(trainer) if graphed else trainer
        for i in range(5):
            stepper.step(*batch, 0.5, i + 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(20):
            stepper.step(*batch, 0.5, i + 6)
        torch.cuda.synchronize()
        times[graphed] = (time.perf_counter() - t0) / 20
    fps = {k: B * (horizon - 2) / v for k, v in times.items()}
    print("configs[0] frames/s: eager %.0f, graphed %.0f" % (fps[False], fps[True]))
    # at this size the eager loop is already GPU-bound on this host (the launches keep ahead of the kernels), so the
    # replay can only tie; it must not be slower by more than noise
    assert times[True] < 1.1 * times[False]
