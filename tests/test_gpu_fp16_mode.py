"""GPU (-m gpu): the opt-in 16-bit MFMA operand mode (ops.set_compute_precision('fp16');
BASELINE configs[4], SURVEY.md 8(d) "cfg5 (fp16 MFMA)").

The reference has no reduced-precision path, so there are two kinds of check:
  * the kernels are EXACT for what they claim: against torch on the same fp16 / bf16
    roundings of the operands they agree to fp32-accumulation accuracy (1e-5 relative);
  * the modules against the reference's fp32 golden vectors at the tolerances SURVEY.md 8(d)
    states for this mode: activations / logits within 1e-2, latent flip rate < 1e-3 per step,
    gradients within 8 % in relative L2 (bf16 operands through up to 6 layers), training-loss trajectory close to the fp32 one.
"""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import weights as WS
from helpers import load, check, to_np

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture()
def fp16_mode():
    from scm_gan_amd import ops
    ops.set_compute_precision("fp16")
    yield ops
    ops.set_compute_precision("fp32")


def _pad(x, circ):
    return F.pad(x, (1, 1, 1, 1), mode="circular") if circ else F.pad(x, (1, 1, 1, 1))


P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("B,c1,c2,cout,H,W,circ", [
    (2, 20, 0, 40, 16, 16, 1), (3, 16, 5, 128, 20, 12, 0), (2, 7, 0, 3, 33, 17, 0), (2, 128, 128, 16, 64, 64, 1),
    (1, 16, 0, 64, 84, 84, 0), (1, 48, 0, 128, 128, 128, 1)])
def test_conv_h16_exact_on_rounded_operands(B, c1, c2, cout, H, W, circ):
    from scm_gan_amd import _lib as L
    lib = L.lib()
    torch.manual_seed(0)
    cin = c1 + c2
    x1 = torch.randn(B, c1, H, W, device="cuda")
    x2 = torch.randn(B, c2, H, W, device="cuda") if c2 else None
    w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    b = torch.randn(cout, device="cuda")
    sc = torch.tensor([0.7], device="cuda")
    fp = torch.empty(lib.scm_conv3x3_pack_h16_bytes(cin, cout) // 2, device="cuda", dtype=torch.int16)
    dp = torch.empty(lib.scm_conv3x3_pack_h16_bytes(cout, cin) // 2, device="cuda", dtype=torch.int16)
    L.check(lib.scm_pack_conv3x3_weights_h16(P(w), cin, cout, 0, 0, 0, P(fp), 1, P(dp), 2, S()), "pack")
    y = torch.empty(B, cout, H, W, device="cuda")
    dy = torch.randn(B, cout, H, W, device="cuda") * 1e-6          # gradient-sized values: below fp16's normal range
    dx1 = torch.empty(B, c1, H, W, device="cuda")
    dx2 = torch.empty(B, c2, H, W, device="cuda") if c2 else None
    d = L.ConvDesc(P(x1), P(x2), c1, c2, None, None, 0, P(fp), cout, P(sc), P(b), None, P(y), None, cout, B, H, W, circ, 1)
    L.check(lib.scm_conv3x3_h16(ctypes.byref(d), 1, S()), "fwd")
    d2 = L.ConvDesc(P(dy), None, cout, 0, P(y), None, 1, P(dp), cin, P(sc), None, None, P(dx1), P(dx2), c1, B, H, W, circ, 0)
    L.check(lib.scm_conv3x3_h16(ctypes.byref(d2), 2, S()), "dgrad")
    x = x1 if x2 is None else torch.cat([x1, x2], 1)
    ref = F.leaky_relu(F.conv2d(_pad(x.half().double(), circ), w.half().double()) * 0.7 + b.double().view(1, -1, 1, 1), 0.01)
    assert ((y.double() - ref).abs().max() / ref.abs().max()).item() < 1e-5
    g = (dy * torch.where(y > 0, torch.ones_like(y), torch.full_like(y, 0.01))).bfloat16().double()
    refd = F.conv2d(_pad(g, circ), w.bfloat16().double().flip(2, 3).transpose(0, 1)) * 0.7
    got = dx1 if dx2 is None else torch.cat([dx1, dx2], 1)
    assert ((got.double() - refd).abs().max() / refd.abs().max()).item() < 1e-5


@pytest.mark.parametrize("B,c1,c2,cout,H,W,circ,gate", [
    (2, 64, 0, 128, 16, 16, 1, 1), (3, 40, 30, 100, 20, 12, 0, 1), (2, 128, 128, 128, 64, 64, 1, 1),
    (1, 64, 0, 128, 84, 84, 0, 0), (1, 48, 0, 128, 128, 128, 1, 2), (2, 24, 0, 128, 16, 16, 1, 1),
    (3, 17, 0, 128, 20, 12, 0, 1), (2, 20, 5, 100, 64, 64, 1, 2)])
def test_wgrad_h16_exact_on_rounded_operands(B, c1, c2, cout, H, W, circ, gate):
    from scm_gan_amd import _lib as L
    lib = L.lib()
    torch.manual_seed(1)
    cin = c1 + c2
    x1 = torch.randn(B, c1, H, W, device="cuda")
    x2 = torch.randn(B, c2, H, W, device="cuda") if c2 else None
    y = torch.rand(B, cout, H, W, device="cuda") if gate == 2 else torch.randn(B, cout, H, W, device="cuda")
    dy = torch.randn(B, cout, H, W, device="cuda") * 1e-5
    pf, cf = ctypes.c_size_t(), ctypes.c_size_t()
    lib.scm_conv3x3_wgrad_workspace(B, H, W, cin, cout, ctypes.byref(pf), ctypes.byref(cf))
    part = torch.empty(pf.value, device="cuda")
    cws = torch.empty(cf.value, device="cuda")
    dw = torch.empty(cout, cin, 3, 3, device="cuda")
    db = torch.empty(cout, device="cuda")
    cs = torch.empty(B, cout, device="cuda")
    d3 = L.WgradDesc(P(x1), P(x2), c1, c2, P(dy), P(y) if gate else None, gate, cout, P(dw), P(db), P(cs), P(part), P(cws),
                     B, H, W, circ, 0, 0, 0, 0, None, None)
    L.check(lib.scm_conv3x3_wgrad_h16(ctypes.byref(d3), 2, S()), "wgrad_h16")
    x = x1 if x2 is None else torch.cat([x1, x2], 1)
    if gate == 1:
        gf = torch.where(y > 0, torch.ones_like(y), torch.full_like(y, 0.01))
    elif gate == 2:
        gf = y * (1 - y)
    else:
        gf = torch.ones_like(y)
    dyp = dy * gf
    xp = _pad(x.bfloat16().double(), circ)
    gr = dyp.bfloat16().double()
    ref = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, device="cuda")
    for ky in range(3):
        for kx in range(3):
            ref[:, :, ky, kx] = torch.einsum("bohw,bihw->oi", gr, xp[:, :, ky:ky + H, kx:kx + W])
    assert ((dw.double() - ref).abs().max() / ref.abs().max()).item() < 1e-5
    cref = dyp.double().sum((2, 3))                  # channel sums are taken before the rounding
    assert ((cs.double() - cref).abs().max() / cref.abs().max()).item() < 1e-5
    assert ((db.double() - cref.sum(0)).abs().max() / cref.sum(0).abs().max()).item() < 1e-5


def _rel(a, e):
    a, e = to_np(a).astype(np.float64), np.asarray(e, dtype=np.float64)
    return np.linalg.norm(a - e) / max(np.linalg.norm(e), 1e-30)


@pytest.mark.parametrize("tag,seed", [("16", 0), ("20x12", 10), ("64", 20)])
def test_modules_fp16_vs_fp32_golden(fp16_mode, tag, seed):
    from scm_gan_amd import models as M
    fx = load("modules_%s.npz" % tag)
    full = "enc_grad.conv2.module.weight_bar" in fx.files or tag != "64"
    enc = M.Encoder(16, 3)
    enc.load_state_dict(WS.to_torch(WS.encoder_state(seed)))
    enc.train()
    z = enc(dev(fx["enc_x"]))
    check(fx, "enc_z", z, 1e-2)
    tr = M.Transition(16, 5)
    tr.load_state_dict(WS.to_torch(WS.transition_state(seed + 1)))
    tr.train()
    zin = dev(fx["tr_z"]).requires_grad_(True)
    a = dev(fx["tr_a"]).requires_grad_(True)
    tr.set_noise([dev(fx["tr_noise"])])
    outs = tr(zin, a, return_all=True)
    for name, o in zip(("skip1", "skip2", "out3", "out4", "out5"), outs):
        check(fx, "tr_" + name, o, 1e-2, 1e-2)
    znext = outs[5].detach().cpu().numpy()
    flips = float((znext != fx["tr_znext"]).mean())
    assert flips < 1e-3, "latent flip rate %.2e" % flips
    if tag != "64":
        (outs[5] * dev(fx["tr_gz"])).sum().backward()
        assert _rel(zin.grad, fx["tr_grad_z"]) < 8e-2
        for k, p in tr.named_parameters():
            if k.endswith("_u") or k.endswith("_v") or "tr_grad." + k not in fx.files:
                continue
            assert _rel(p.grad, fx["tr_grad." + k]) < 8e-2, k
    dec = M.Decoder(16, 3)
    dec.load_state_dict(WS.to_torch(WS.decoder_state(seed + 2)))
    y = dec(dev(fx["dec_z"]))
    check(fx, "dec_y", y, 1e-2, 1e-2)
    rp = M.RewardPredictor(16, 2)
    rp.load_state_dict(WS.to_torch(WS.reward_state(seed + 3)))
    r = rp(dev(fx["rp_z"]))
    check(fx, "rp_r", r, 1e-2, 1e-2)


def test_training_fp16_tracks_fp32(fp16_mode):
    """Five optimizer steps from the same weights, batches and latent noise in both modes:
    the loss trajectories agree to a few % and both decrease."""
    from scm_gan_amd import main as T, models as M, ops
    from scm_gan_amd.datasource import SyntheticDatasource

    def run(mode):
        ops.set_compute_precision(mode)
        torch.manual_seed(3)
        enc, dec, rp, disc, tr = M.Encoder(16, 3), M.Decoder(16, 3), M.RewardPredictor(16, 2), M.Discriminator(), \
            M.Transition(16, 5)
        tr.seed_noise(7)
        opts = T.parser.parse_args(["--horizon-min", "5", "--horizon-max", "5", "--batch-size", "8"])
        trainer = T.Trainer(16, 5, 2, enc, dec, rp, disc, tr, opts)
        ds = SyntheticDatasource(seed=5, height=32, width=32, channels=3, num_actions=5, num_rewards=2)
        losses = []
        for i in range(5):
            s, r, d, a = ds.get_trajectories(8, 5)[:4]
            st = torch.as_tensor(s, dtype=torch.float32).cuda()
            out = trainer.step(st, torch.as_tensor(r, dtype=torch.float32).cuda(),
                               torch.as_tensor(d.astype(np.float32)).cuda(), T.onehot(a, 5), theta=0.5, train_iter=i + 1)
            losses.append(float(out))
        return np.array(losses)

    l16 = run("fp16")
    l32 = run("fp32")
    assert np.all(np.isfinite(l16))
    assert np.max(np.abs(l16 - l32) / np.abs(l32)) < 5e-2, (l16, l32)
    assert l16[-1] < l16[0]


def test_cfg5_shape_step_fp16(fp16_mode):
    """BASELINE configs[4] shape (128x128x4 frames, 4 actions, 2 rewards; small batch/horizon here):
    one optimizer step in the 16-bit mode against the fp32 path from the same state."""
    from scm_gan_amd import main as T, models as M, ops
    from scm_gan_amd.datasource import SyntheticDatasource

    def run(mode):
        ops.set_compute_precision(mode)
        torch.manual_seed(11)
        enc, dec, rp, disc, tr = M.Encoder(16, 4), M.Decoder(16, 4), M.RewardPredictor(16, 2), M.Discriminator(), \
            M.Transition(16, 4)
        tr.seed_noise(3)
        opts = T.parser.parse_args(["--horizon-min", "4", "--horizon-max", "4", "--batch-size", "2"])
        trainer = T.Trainer(16, 4, 2, enc, dec, rp, disc, tr, opts)
        ds = SyntheticDatasource(seed=9, height=128, width=128, channels=4, num_actions=4, num_rewards=2)
        s, r, d, a = ds.get_trajectories(2, 4)[:4]
        loss = trainer.step(torch.as_tensor(s, dtype=torch.float32).cuda(), torch.as_tensor(r, dtype=torch.float32).cuda(),
                            torch.as_tensor(d.astype(np.float32)).cuda(), T.onehot(a, 4), theta=0.5, train_iter=1)
        w = torch.cat([p.detach().flatten() for p in list(tr.parameters()) + list(enc.parameters())])
        return float(loss), w

    l16, w16 = run("fp16")
    l32, w32 = run("fp32")
    assert abs(l16 - l32) / abs(l32) < 2e-2, (l16, l32)
    # Adam's first step moves every weight by ~lr * sign(g): the two modes agree wherever the gradient sign does
    assert float((w16 - w32).abs().max()) < 3e-3


def test_half_stored_activations_equal_fp32_tensors(fp16_mode):
    """16-bit operand mode: with the hidden maps of the encoder / transition STORED as IEEE half (ops.half_store_ok:
    y_half / x_half of scm_conv3x3_h16, gradients through cells) the forward is bitwise the one over fp32 tensors --
    the operands were rounded to fp16 once either way -- and the gradients agree to the double rounding of the
    weight gradient's bf16 operands (bf16(fp16(x)) vs bf16(x)); the half tensors really are what crosses HBM."""
    from scm_gan_amd import main as T, models as M, ops
    from scm_gan_amd.datasource import SyntheticDatasource

    def run(half):
        ops.set_half_store(half)
        torch.manual_seed(5)
        enc, dec, rp, disc, tr = M.Encoder(16, 3), M.Decoder(16, 3), M.RewardPredictor(16, 2), M.Discriminator(), \
            M.Transition(16, 5)
        tr.seed_noise(11)
        seen = []
        hooks = [m.register_forward_hook(lambda mod, i, o: seen.append(o.dtype)) for m in (tr.conv2, tr.conv5, tr.conv6, enc.conv2)]
        opts = T.parser.parse_args(["--horizon-min", "5", "--horizon-max", "5", "--batch-size", "4"])
        trainer = T.Trainer(16, 5, 2, enc, dec, rp, disc, tr, opts)
        ds = SyntheticDatasource(seed=8, height=32, width=32, channels=3, num_actions=5, num_rewards=2, done_every=3)
        s, r, d, a = ds.get_trajectories(4, 5)[:4]
        loss = trainer.step(torch.as_tensor(s, dtype=torch.float32).cuda(), torch.as_tensor(r, dtype=torch.float32).cuda(),
                            torch.as_tensor(d.astype(np.float32)).cuda(), T.onehot(a, 5), theta=0.5, train_iter=1)
        for h in hooks:
            h.remove()
        grads = {n + "." + k: p.grad.detach().clone() for n, m in (("enc", enc), ("tr", tr), ("dec", dec), ("rp", rp))
                 for k, p in m.named_parameters() if p.grad is not None}
        return float(loss), grads, seen

    try:
        l_h, g_h, dt_h = run(True)
        l_f, g_f, dt_f = run(False)
    finally:
        ops.set_half_store(True)
    assert torch.float16 in dt_h and torch.float32 in dt_h          # conv2 / enc.conv2 half; conv5 / conv6 fp32
    assert set(dt_f) == {torch.float32}
    assert l_h == l_f                                               # the forward is bitwise the same
    assert g_h.keys() == g_f.keys()
    for k in g_f:
        den = float(g_f[k].norm()) + 1e-12
        assert float((g_h[k] - g_f[k]).norm()) / den < 2e-3, k
