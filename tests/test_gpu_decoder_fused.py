"""GPU (-m gpu): the fused decoder + pixel-loss kernels (scm_decoder_fused_fwd / _bwd) through the C ABI.

  * forward logits against the reference's own Decoder outputs (tests/golden/modules_*.npz, dec_y), 1e-4;
  * loss, dz and the four ConvTranspose2d parameter gradients against the CPU oracle (models.py:270-291 +
    main.py:188-190 / 310-312 and 814-816) on ragged, narrow, wide (x-tiled) and multi-band shapes;
  * full configs[1] size: fused == layer-by-layer HIP path (the hidden map's leaky-relu slope may differ where a
    pre-activation is within rounding of zero: counted, not ignored);
  * gradients accumulate into existing .grad buffers under ops.grad_sinks (Trainer's flat buckets);
  * C = 4 (sc2-shaped frames) keeps the layer-by-layer path.
"""
import numpy as np
import pytest
import torch

import weights as WS
from helpers import load, check

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def make_decoder(seed, C=3):
    from scm_gan_amd import models as M
    dec = M.Decoder(16, C)
    dec.load_state_dict(WS.to_torch(WS.decoder_state(seed, color=C)))
    return dec


@pytest.mark.parametrize("tag,seed", [("16", 0), ("20x12", 10), ("64", 20)])
def test_fused_forward_vs_reference_fixture(tag, seed):
    fx = load("modules_%s.npz" % tag)
    dec = make_decoder(seed + 2)
    assert dec._fused()
    with torch.no_grad():
        y = dec(dev(fx["dec_z"]))                      # grad disabled -> scm_decoder_fused_fwd
    check(fx, "dec_y", y, 1e-4)
    # the layer-by-layer path (grad enabled) gives the same logits
    y2 = dec(dev(fx["dec_z"]))
    assert float((y - y2.detach()).abs().max()) < 2e-5


SHAPES = [(2, 16, 16, 3, 0), (1, 20, 12, 3, 0), (3, 7, 13, 1, 0), (2, 33, 20, 2, 1), (2, 48, 64, 3, 0), (1, 84, 84, 3, 0),
          (1, 70, 130, 2, 1), (5, 64, 64, 3, 0)]


@pytest.mark.parametrize("B,H,W,C,kind", SHAPES)
def test_fused_loss_and_gradients_vs_oracle(B, H, W, C, kind):
    from oracle import scm_oracle as O
    seed = 7 * H + W
    rng = np.random.default_rng(seed)
    z = (rng.random((B, 16, H, W)) < 0.5).astype(np.float32) * rng.random((B, 16, H, W)).astype(np.float32)
    tgt = (rng.random((B, C, H, W)) < 0.2).astype(np.float32)
    gl = (rng.random(B) + 0.5).astype(np.float32)
    # ---- oracle
    ost = O.clone_state(WS.decoder_state(seed, color=C))
    for k in ost:
        ost[k].requires_grad_(True)
    oz = torch.from_numpy(z).requires_grad_(True)
    ol = O.decoder_forward(ost, oz)
    op = torch.sigmoid(ol)
    oloss = O.decoder_pixel_loss(torch.from_numpy(tgt), op) if kind == 0 else \
        ((torch.from_numpy(tgt) - op) ** 2).mean(-1).mean(-1).mean(-1)
    (oloss * torch.from_numpy(gl)).sum().backward()
    # ---- fused HIP
    dec = make_decoder(seed, C)
    hz = dev(z).requires_grad_(True)
    loss = dec.forward_loss(hz, dev(tgt), kind)
    (loss * dev(gl)).sum().backward()
    assert float((loss.detach().cpu() - oloss.detach()).abs().max()) < 1e-5
    pairs = [("dz", hz.grad, oz.grad)] + [(k, p.grad, ost[k].grad) for k, p in dec.named_parameters()]
    for name, got, ref in pairs:
        got, ref = got.cpu(), ref
        scale = max(1e-7, float(ref.abs().max()))
        err = (got - ref).abs()
        # a hidden unit whose pre-activation is within fp32 rounding of 0 may take the other leaky-relu slope
        bad = int((err > 1e-4 * scale + 1e-9).sum())
        assert bad <= got.numel() // 20000, (name, float(err.max()), scale, bad)
        assert float(err.max()) < 5e-2 * scale, (name, float(err.max()), scale)


def test_fused_equals_layer_path_at_configs1_size():
    from scm_gan_amd import ops
    B, H, W, C = 64, 64, 64, 3
    torch.manual_seed(3)
    dec = make_decoder(5)
    z = (torch.rand(B, 16, H, W, device="cuda") < 0.5).float().requires_grad_(True)
    tgt = (torch.rand(B, C, H, W, device="cuda") < 0.1).float()
    x = dec.conv1(z, act=ops.ACT_LRELU)
    ref_loss = ops.PixelLossFn.apply(dec.conv2.forward_factor_sum(x, 16, C), tgt, 0)
    ref_loss.mean().backward()
    ref = [z.grad.clone()] + [p.grad.clone() for p in dec.parameters()]
    z.grad = None
    for p in dec.parameters():
        p.grad = None
    loss = dec.forward_loss(z, tgt)
    loss.mean().backward()
    got = [z.grad] + [p.grad for p in dec.parameters()]
    assert float((loss - ref_loss).abs().max()) < 2e-6
    flips = 0
    for g, r in zip(got, ref):
        scale = float(r.abs().max())
        bad = int(((g - r).abs() > 1e-4 * scale).sum())
        flips += bad
        assert bad <= g.numel() // 20000 + 2, (bad, g.numel())
    # run twice: bitwise reproducible (fixed-order partial sums)
    z.grad = None
    for p in dec.parameters():
        p.grad = None
    dec.forward_loss(z, tgt).mean().backward()
    for g, h in zip(got, [z.grad] + [p.grad for p in dec.parameters()]):
        assert torch.equal(g, h)


def test_fused_backward_accumulates_into_grad_sinks():
    from scm_gan_amd import ops
    dec = make_decoder(9)
    z = (torch.rand(2, 16, 16, 16, device="cuda") < 0.5).float()
    tgt = (torch.rand(2, 3, 16, 16, device="cuda") < 0.2).float()
    dec.forward_loss(z, tgt).sum().backward()
    once = [p.grad.clone() for p in dec.parameters()]
    with ops.grad_sinks():
        dec.forward_loss(z, tgt).sum().backward()      # adds into the existing .grad in-kernel
    for p, g in zip(dec.parameters(), once):
        assert torch.allclose(p.grad, 2 * g, rtol=1e-6, atol=1e-9)


def test_four_channel_frames_take_the_layer_path():
    dec = make_decoder(11, C=4)
    assert not dec._fused()
    z = (torch.rand(1, 16, 12, 12, device="cuda") < 0.5).float()
    tgt = (torch.rand(1, 4, 12, 12, device="cuda") < 0.2).float()
    from scm_gan_amd import ops
    with torch.no_grad():
        a = dec.forward_loss(z, tgt)
        b = ops.pixel_bce(dec(z), tgt)
    assert torch.equal(a, b)


# ------------------------------------------------------------------ fused reward-head convolutions
REWARD_SHAPES = [(2, 16, 16, 2), (1, 20, 12, 2), (3, 7, 13, 1), (2, 33, 20, 2), (2, 48, 64, 2), (1, 5, 5, 2), (2, 64, 31, 1),
                 (4, 64, 64, 2)]


@pytest.mark.parametrize("B,H,W,R", REWARD_SHAPES)
def test_fused_reward_head_vs_oracle(B, H, W, R):
    """scm_reward_fused_fwd / _bwd (RewardPredictor.conv1 + leaky_relu + conv2 in one pass, models.py:235-240) + the
    softmax tail against the CPU oracle: r, the per-position map, dz and the four Conv2d gradients."""
    from oracle import scm_oracle as O
    from scm_gan_amd import models as M
    seed = 11 * H + W
    rng = np.random.default_rng(seed)
    z = (rng.random((B, 16, H, W)) < 0.5).astype(np.float32)
    gr = rng.standard_normal((B, R)).astype(np.float32)
    ost = O.clone_state(WS.reward_state(seed, rewards=R))
    for k in ost:
        ost[k].requires_grad_(True)
    oz = torch.from_numpy(z).requires_grad_(True)
    orr, omap = O.reward_forward(ost, oz, visualize=True)
    ((orr * torch.from_numpy(gr)).sum() + 0.01 * omap.sum()).backward()
    rp = M.RewardPredictor(16, R)
    rp.load_state_dict(WS.to_torch(WS.reward_state(seed, rewards=R)))
    hz = dev(z).requires_grad_(True)
    assert rp._fused(hz)
    r, rmap = rp(hz, visualize=True)
    ((r * dev(gr)).sum() + 0.01 * rmap.sum()).backward()
    assert float((r.detach().cpu() - orr.detach()).abs().max()) < 1e-4 * max(1.0, float(orr.abs().max()))
    assert float((rmap.detach().cpu() - omap.detach()).abs().max()) < 1e-5
    pairs = [("dz", hz.grad, oz.grad)] + [(k, p.grad, ost[k].grad) for k, p in rp.named_parameters()]
    for name, got, ref in pairs:
        scale = max(1e-7, float(ref.abs().max()))
        err = (got.cpu() - ref).abs()
        bad = int((err > 1e-4 * scale + 1e-9).sum())
        assert bad <= got.numel() // 20000, (name, float(err.max()), scale, bad)


def test_wide_frames_keep_the_layer_path_for_the_reward_head():
    from scm_gan_amd import models as M
    rp = M.RewardPredictor(16, 1)
    z = (torch.rand(1, 16, 20, 84, device="cuda") < 0.5).float()
    assert not rp._fused(z)
    assert rp(z).shape == (1, 1)


@pytest.mark.parametrize("B,H,W,C", [(3, 64, 64, 3), (2, 84, 84, 3), (2, 33, 20, 2), (9, 16, 16, 1)])
def test_kept_hidden_map_gives_bitwise_the_recomputed_gradients(B, H, W, C):
    """scm_decoder_fused_fwd's h_save / scm_decoder_fused_bwd's h_saved and the reward head's: the backward
    kernels that LOAD the hidden map the forward kept produce bitwise the gradients of the ones that recompute it from z
    (same MFMA sequence in the forward and in the recomputation), on single-tile, x-tiled (W > 64), ragged and multi-band
    shapes."""
    from scm_gan_amd import models as M, ops
    rng = np.random.default_rng(5 * H + W)
    zn = ((rng.random((B, 16, H, W)) < 0.5) * rng.random((B, 16, H, W))).astype(np.float32)
    tgt = dev((rng.random((B, C, H, W)) < 0.2).astype(np.float32))
    gl = dev((rng.random(B) + 0.5).astype(np.float32))
    dec = make_decoder(H + W, C)
    rp = M.RewardPredictor(16, 2)
    rp.load_state_dict(WS.to_torch(WS.reward_state(H + W + 1, rewards=2)))
    assert dec._fused()
    fused_rp = rp._fused(dev(zn))
    out = {}
    old = ops.save_hidden
    try:
        for keep in (True, False):
            ops.save_hidden = keep
            res = []
            for net, run in ((dec, lambda z: (dec.forward_loss(z, tgt) * gl).sum()),
                             (rp, lambda z: (rp(z) * gl[:, None]).sum())):
                z = dev(zn).requires_grad_(True)
                for p in net.parameters():
                    p.grad = None
                run(z).backward()
                res.append([z.grad.clone()] + [p.grad.clone() for p in net.parameters()])
            out[keep] = res
    finally:
        ops.save_hidden = old
    for a, b in zip(out[True], out[False]):
        for ga, gb in zip(a, b):
            assert torch.equal(ga, gb)
    assert fused_rp == (W <= 64 and H >= 5 and W >= 5)
