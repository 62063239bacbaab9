"""Time-batched heads (scm_gan_amd.main._heads_time_batched): the reward head and the decoder + pixel loss run once
over all unrolled steps instead of once per step (reference main.py:177-197 issues them per step).  Same per-sample
arithmetic, so against the per-step launches of the SAME build: per-step logged losses and the total loss bitwise,
gradients to summation-order rounding (the heads' weight gradients are reduced over all steps' workgroups at once).
The reference / oracle comparisons of the batched path are the training-step tests of test_gpu_parity.py,
test_gpu_round2.py, test_gpu_properties.py and test_gpu_configs34.py, which run it by default."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import weights as WS  # noqa: E402

pytestmark = pytest.mark.gpu


def _run(mode, shape, flags=()):
    from scm_gan_amd import models as M, main as T
    from scm_gan_amd.datasource import SyntheticDatasource
    H, W, C, A, R, B, horizon = shape
    L, seed = 16, 40 + H
    old = os.environ.get("SCM_BATCHED_HEADS")
    os.environ["SCM_BATCHED_HEADS"] = mode
    try:
        states = [WS.encoder_state(seed, color=C), WS.transition_state(seed + 1, actions=A),
                  WS.decoder_state(seed + 2, color=C), WS.reward_state(seed + 3, rewards=R)]
        ds = SyntheticDatasource(seed=seed, height=H, width=W, channels=C, num_actions=A, num_rewards=R, density=0.3,
                                 done_every=3)
        s, r, d, a = ds.get_trajectories(B, horizon)
        rng = np.random.default_rng(seed)
        noise = [rng.random((B, L, H, W), dtype=np.float32) for _ in range(horizon - 2)]
        enc, tr, dec, rp = M.Encoder(L, C), M.Transition(L, A), M.Decoder(L, C), M.RewardPredictor(L, R)
        for m, st in zip((enc, tr, dec, rp), states):
            m.load_state_dict(WS.to_torch(st))
        opts = T.parser.parse_args(["--batch-size", str(B), "--train-iters", "2"] + list(flags))
        trainer = T.Trainer(L, A, R, enc, dec, rp, M.Discriminator(), tr, opts)
        tr.set_noise([torch.from_numpy(n).cuda() for n in noise])
        up = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
        loss = trainer.step(up(s), up(r), up(d.astype(np.float32)), T.onehot(a, A), 0.5, 1, a)
        logs = {k: float(v[-1]) for k, v in trainer.ts.series.items() if v}
        grads = {f.name: f.flat.clone() for f in trainer.flats}          # (the clamped full-batch gradients)
        return float(loss), logs, grads
    finally:
        if old is None:
            os.environ.pop("SCM_BATCHED_HEADS", None)
        else:
            os.environ["SCM_BATCHED_HEADS"] = old


@pytest.mark.parametrize("shape,flags", [
    ((64, 64, 3, 5, 2, 8, 6), ()),                      # fused decoder + fused reward head (the headline's kernels)
    ((84, 84, 3, 4, 1, 2, 5), ()),                      # fused decoder (overlapping column tiles), layer-by-layer reward head
    ((32, 32, 4, 4, 2, 3, 5), ()),                      # C = 4: layer-by-layer decoder
    ((16, 16, 3, 5, 2, 4, 6), ("--truncate-bptt",)),    # the heads read the latent from before the detach
])
def test_time_batched_heads_equal_per_step_launches(shape, flags):
    loss_b, logs_b, grads_b = _run("", shape, flags)        # default: batched for this package's modules
    loss_s, logs_s, grads_s = _run("0", shape, flags)       # per-step launches
    nT = shape[6] - 2
    assert len([k for k in logs_b if k.startswith("Reconstruction t=")]) == nT
    assert logs_b.keys() == logs_s.keys()
    for k in logs_s:
        assert logs_b[k] == logs_s[k], (k, logs_b[k], logs_s[k])
    assert loss_b == loss_s
    for name in grads_s:
        gb, gs = grads_b[name], grads_s[name]
        scale = float(gs.abs().max())
        assert float((gb - gs).abs().max()) <= 2e-5 * max(scale, 1e-6), (name, float((gb - gs).abs().max()), scale)


def test_latent_overshooting_keeps_the_per_step_heads():
    """--latent-overshooting reads the running mask inside the loop (main.py:217-230): not batched."""
    from scm_gan_amd import main as T
    opts = T.parser.parse_args(["--latent-overshooting"])
    z = torch.zeros(2, 16, 16, 16)

    class Own(object):
        def _fused(self, *a):
            return True
    assert not T._heads_time_batched(opts, Own(), Own(), z, 8)
    assert T._heads_time_batched(T.parser.parse_args([]), Own(), Own(), z, 8)
    assert not T._heads_time_batched(T.parser.parse_args([]), Own(), Own(), z, 1)
    assert not T._heads_time_batched(T.parser.parse_args([]), object(), object(), z, 8)
