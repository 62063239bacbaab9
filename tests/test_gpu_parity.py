"""GPU (-m gpu): the HIP path, called through the C ABI, against the golden
vectors produced by the reference's own modules (tests/golden/make_golden.py)
and against the CPU oracle.  Tolerance: fp32, 1e-4 per element on activations
and decoded pixels (BASELINE.json north_star), binary latents bit-exact."""
import numpy as np
import pytest
import torch

import weights as WS
from helpers import load, check

pytestmark = pytest.mark.gpu

ATOL = 1e-4


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def M():
    from scm_gan_amd import models
    assert torch.cuda.is_available()
    return models


@pytest.mark.parametrize("tag,seed,grads", [("16", 0, True), ("20x12", 10, True), ("64", 20, False)])
def test_modules_vs_golden(M, tag, seed, grads):
    fx = load("modules_%s.npz" % tag)
    # ---- encoder
    enc = M.Encoder(16, 3)
    enc.load_state_dict(WS.to_torch(WS.encoder_state(seed)))
    enc.train()
    x = dev(fx["enc_x"])
    z = enc(x)
    check(fx, "enc_z", z, ATOL)
    if grads:
        (z * dev(fx["enc_gz"])).sum().backward()
        for k, p in enc.named_parameters():
            if k.startswith("bn_") or k.endswith("_u") or k.endswith("_v"):
                assert p.grad is None
                continue
            check(fx, "enc_grad." + k, p.grad, 2e-4, 1e-3, "encoder")
    sd = enc.state_dict()
    for k in sd:
        if k.endswith("weight_u") or k.endswith("weight_v"):
            check(fx, "enc_after1." + k, sd[k], 1e-5)
    check(fx, "enc_z_call2", enc(x), ATOL)

    # ---- transition, train mode with the recorded noise
    tr = M.Transition(16, 5)
    tr.load_state_dict(WS.to_torch(WS.transition_state(seed + 1)))
    tr.train()
    zin = dev(fx["tr_z"]).requires_grad_(True)
    a = dev(fx["tr_a"]).requires_grad_(True)
    tr.set_noise([dev(fx["tr_noise"])])
    outs = tr(zin, a, return_all=True)
    for name, o in zip(("skip1", "skip2", "out3", "out4", "out5"), outs):
        check(fx, "tr_" + name, o, ATOL)
    # binary latent: identical wherever the reference logit is not within 1e-5 of the noise threshold
    znext = outs[5].detach().cpu().numpy()
    ref = fx["tr_znext"]
    p_ref = 1.0 / (1.0 + np.exp(-fx["tr_logits"].astype(np.float64)))
    safe = np.abs(p_ref - fx["tr_noise"]) > 1e-5
    assert np.array_equal(znext[safe], ref[safe])
    assert (znext != ref).sum() <= 2
    if grads:
        (outs[5] * dev(fx["tr_gz"])).sum().backward()
        check(fx, "tr_grad_z", zin.grad, 2e-4, 1e-3)
        check(fx, "tr_grad_a", a.grad, 2e-3, 1e-3)
        for k, p in tr.named_parameters():
            if k.endswith("_u") or k.endswith("_v"):
                continue
            check(fx, "tr_grad." + k, p.grad, 5e-4, 1e-3, "transition")
    sd = tr.state_dict()
    for k in sd:
        if k.endswith("weight_u") or k.endswith("weight_v"):
            check(fx, "tr_after1." + k, sd[k], 1e-5)
    tr.eval()
    with torch.no_grad():
        zn = tr(dev(fx["tr_eval_z"]), dev(fx["tr_a"]))
    safe = np.abs(fx["tr_eval_logits"]) > 1e-5
    assert np.array_equal(zn.cpu().numpy()[safe], fx["tr_eval_znext"][safe])

    # ---- decoder
    dec = M.Decoder(16, 3)
    dec.load_state_dict(WS.to_torch(WS.decoder_state(seed + 2)))
    zd = dev(fx["dec_z"]).requires_grad_(True)
    y, vis = dec(zd, visualize=True)
    check(fx, "dec_y", y, ATOL)
    check(fx, "dec_vis", vis, ATOL)
    if grads:
        (y * dev(fx["dec_gy"])).sum().backward()
        check(fx, "dec_grad_z", zd.grad, 2e-4, 1e-3)
        for k, p in dec.named_parameters():
            check(fx, "dec_grad." + k, p.grad, 1e-3, 1e-3, "decoder")

    # ---- reward predictor
    rp = M.RewardPredictor(16, 2)
    rp.load_state_dict(WS.to_torch(WS.reward_state(seed + 3)))
    zr = dev(fx["rp_z"]).requires_grad_(True)
    r, rmap = rp(zr, visualize=True)
    check(fx, "rp_r", r, 2e-4, 1e-4)
    check(fx, "rp_map", rmap, 1e-5)
    if grads:
        (r * dev(fx["rp_gr"])).sum().backward()
        check(fx, "rp_grad_z", zr.grad, 1e-5, 1e-3)
        for k, p in rp.named_parameters():
            check(fx, "rp_grad." + k, p.grad, 2e-4, 1e-3, "reward")


def test_operators_vs_golden(M):
    from scm_gan_amd.layers import Conv2d
    from scm_gan_amd.spectral_normalization import SpectralNorm
    from scm_gan_amd.coordconv import CoordConv2d
    fx = load("operators.npz")
    # ---- SpectralNorm call sequence
    sn = SpectralNorm(Conv2d(128, 128, 3, padding=1)).cuda()
    st = WS.transition_state(0)
    sn.load_state_dict({"module." + k.split(".module.")[1]: torch.from_numpy(v)
                        for k, v in st.items() if k.startswith("conv2.module.")})
    x = dev(fx["sn_x"])
    for call in (1, 2, 3):
        y = sn(x)
        check(fx, "sn_u%d" % call, sn.module.weight_u, 1e-5)
        check(fx, "sn_v%d" % call, sn.module.weight_v, 1e-5)
        check(fx, "sn_y%d" % call, y, ATOL)
    (y * dev(fx["sn_gy"])).sum().backward()
    check(fx, "sn_grad_wbar", sn.module.weight_bar.grad, 2e-4, 1e-3)
    check(fx, "sn_grad_bias", sn.module.bias.grad, 2e-4, 1e-4)

    # ---- Discriminator (documented flatten): logits within 1e-3 (north_star)
    disc = M.Discriminator()
    disc.load_state_dict(WS.to_torch(WS.discriminator_state(4)))
    xd = dev(fx["disc_x"]).requires_grad_(True)
    logit = disc(xd)
    check(fx, "disc_logit", logit, 1e-3)
    check(fx, "disc_logit", logit, 1e-4)
    (logit * dev(fx["disc_gl"])).sum().backward()
    check(fx, "disc_grad_x", xd.grad, 1e-5, 1e-3)
    for k, p in disc.named_parameters():
        if k.endswith("_u") or k.endswith("_v"):
            continue
        check(fx, "disc_grad." + k, p.grad, 2e-4, 1e-3, "discriminator")

    # ---- CoordConv2d
    cc = CoordConv2d(6 + 2, 10, 3, padding=1).cuda()
    cc.load_state_dict(WS.to_torch(WS.coordconv_state(5, 6, 10)))
    xc = dev(fx["cc_x"]).requires_grad_(True)
    yc = cc(xc)
    check(fx, "cc_y", yc, ATOL)
    (yc * dev(fx["cc_gy"])).sum().backward()
    check(fx, "cc_grad_x", xc.grad, ATOL)
    check(fx, "cc_grad.conv.weight", cc.conv.weight.grad, 5e-4, 1e-4)
