"""GPU (-m gpu): the Winograd F(2x2, 3x3) form of the wide fp32 convolutions (csrc/conv3x3_wino.hip,
conv3x3_wgrad_wino.hip; the default algorithm, ops.set_conv_algo) -- forward, gated input gradient and weight / bias
gradient through the C ABI against FLOAT64, next to the direct implicit-GEMM kernels on the same data.

What is claimed and tested: it is the same fp32 operator (fp32 operands, fp32 MFMA accumulation, transforms with
entries 0, +-1, +-1/2), so its error against float64 is of the direct kernels' size (measured 0.3x .. 1.5x: it sums
fewer products), the 1-bit gate it writes is the sign of its own output, the epilogue contract (scale, bias,
per-sample bias, split outputs, accumulate) is the direct kernel's, results are bitwise reproducible, and frames it
does not take (odd widths) silently run the direct kernels.  The reference-generated golden tests and the full-size
property tests run in this algorithm by default; `SCM_CONV_ALGO=direct` re-runs the suite on the direct kernels."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [  # B, c1, c2, cout, H, W, circular, act
    (2, 32, 0, 64, 16, 16, 1, 1), (2, 128, 128, 128, 16, 16, 1, 1), (2, 9, 0, 128, 16, 16, 0, 1),
    (1, 16, 0, 64, 20, 12, 0, 1), (3, 64, 0, 48, 20, 12, 0, 0), (1, 128, 0, 128, 64, 64, 1, 1),
    (1, 21, 0, 128, 84, 84, 1, 1), (1, 8, 0, 256, 16, 16, 1, 0), (1, 128, 0, 128, 128, 128, 0, 1),
    (2, 40, 0, 70, 9, 8, 1, 1), (2, 40, 0, 70, 9, 8, 0, 2), (1, 24, 0, 64, 33, 22, 1, 0), (2, 16, 8, 100, 6, 6, 0, 1),
    (1, 12, 0, 64, 2, 2, 1, 0), (1, 12, 0, 64, 3, 6, 0, 0), (2, 128, 0, 128, 32, 32, 1, 1), (1, 64, 0, 64, 10, 100, 0, 1),
    # 128-channel workgroups on runs of 32 patches that straddle patch rows, with a ragged last workgroup: 84 x 84
    # (BASELINE configs[3]), odd heights, a 6-patch row; 128-wide frames with two input tensors
    (2, 128, 0, 128, 84, 84, 1, 1), (1, 128, 128, 128, 84, 84, 0, 1), (2, 32, 0, 128, 33, 22, 1, 1), (3, 64, 0, 128, 20, 12, 0, 0),
    (1, 128, 128, 256, 128, 128, 1, 1)]
WSHAPES = [
    (2, 64, 0, 64, 16, 16, 1, 1), (2, 128, 128, 128, 16, 16, 1, 1), (2, 128, 0, 128, 16, 16, 0, 1),
    (1, 64, 0, 64, 20, 12, 0, 1), (3, 64, 0, 48, 20, 12, 0, 0), (1, 128, 0, 128, 64, 64, 1, 1),
    (1, 100, 0, 128, 84, 84, 1, 1), (1, 128, 0, 128, 128, 128, 0, 1), (2, 70, 0, 70, 9, 8, 1, 1),
    (2, 70, 0, 70, 9, 8, 0, 2), (1, 64, 0, 64, 33, 24, 1, 0), (2, 64, 64, 100, 6, 8, 0, 1),
    (1, 64, 0, 64, 2, 4, 1, 0), (4, 128, 0, 128, 32, 32, 1, 1), (1, 64, 0, 64, 10, 100, 0, 1),
    # bands of several patch rows on frames with W % 32 == 0: the zigzag form (conv3x3_wgrad_winoz.hip) -- carried rows,
    # role swaps, strip pairs, zero padding at the top / bottom band, two input tensors, ungated / float / 1-bit gates
    (8, 128, 0, 128, 64, 64, 1, 1), (8, 64, 64, 128, 64, 64, 0, 1), (64, 64, 0, 64, 32, 32, 1, 0),
    (3, 128, 0, 128, 128, 128, 0, 1), (2, 64, 0, 64, 64, 32, 0, 0),
    # ... with a ragged last strip (W % 32 != 0, W % 4 == 0): 84 x 84 (BASELINE configs[3]) in bands of 6 and 2 patch rows, 20 x 52
    (8, 128, 0, 128, 84, 84, 1, 1), (4, 64, 64, 64, 84, 84, 0, 1), (4, 64, 0, 64, 20, 52, 1, 0)]


@pytest.fixture(scope="module")
def K():
    from scm_gan_amd import _lib as L

    class Kit(object):
        lib = L.lib()
        P = staticmethod(lambda t: None if t is None else ctypes.c_void_p(t.data_ptr()))
        S = staticmethod(lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))

        def pack(self, w, wino):
            cout, cin = w.shape[:2]
            size = self.lib.scm_conv3x3_pack_wino_floats if wino else self.lib.scm_conv3x3_pack_floats
            fn = self.lib.scm_pack_conv3x3_weights_wino if wino else self.lib.scm_pack_conv3x3_weights
            fp, dp = torch.empty(size(cin, cout), device="cuda"), torch.empty(size(cout, cin), device="cuda")
            L.check(fn(self.P(w), cin, cout, 0, 0, 0, self.P(fp), self.P(dp), self.S()), "pack")
            return fp, dp

        def conv(self, wino, x1, x2, wp, cout, pad, act, bias=None, sbias=None, scale=None, gate1=None, gate_act=0,
                 co1=None, mask_out=None, mask_in=None, y1_acc=None):
            B, c1, H, W = x1.shape
            c2 = 0 if x2 is None else x2.shape[1]
            co1 = cout if co1 is None else co1
            y1 = torch.empty(B, co1, H, W, device="cuda") if y1_acc is None else y1_acc
            y2 = torch.empty(B, cout - co1, H, W, device="cuda") if co1 < cout else None
            d = L.ConvDesc(self.P(x1), self.P(x2), c1, c2, self.P(gate1), None, gate_act, self.P(wp), cout, self.P(scale),
                           self.P(bias), self.P(sbias), self.P(y1), self.P(y2), co1, B, H, W, pad, act, self.P(mask_out),
                           self.P(mask_in), 0 if y1_acc is None else 1)
            L.check((self.lib.scm_conv3x3_wino if wino else self.lib.scm_conv3x3)(ctypes.byref(d), self.S()), "conv")
            return y1, y2

        def wgrad(self, wino, x1, x2, dy, gate, gate_act, pad, mask=None):
            B, c1, H, W = x1.shape
            c2 = 0 if x2 is None else x2.shape[1]
            cout, cin = dy.shape[1], c1 + c2
            pf, cf = ctypes.c_size_t(), ctypes.c_size_t()
            L.check(self.lib.scm_conv3x3_wgrad_workspace(B, H, W, cin, cout, ctypes.byref(pf), ctypes.byref(cf)), "ws")
            part, cs = torch.empty(pf.value, device="cuda"), torch.empty(cf.value, device="cuda")
            dw, db, csum = torch.empty(cout, cin, 3, 3, device="cuda"), torch.empty(cout, device="cuda"), torch.empty(B, cout, device="cuda")
            d = L.WgradDesc(self.P(x1), self.P(x2), c1, c2, self.P(dy), self.P(gate), gate_act, cout, self.P(dw), self.P(db),
                            self.P(csum), self.P(part), self.P(cs), B, H, W, pad, 0, 0, 0, 0, None, None, self.P(mask))
            L.check((self.lib.scm_conv3x3_wgrad_wino if wino else self.lib.scm_conv3x3_wgrad)(ctypes.byref(d), self.S()), "wgrad")
            return dw, db, csum
    return Kit()


def ref_conv(x, w, b, pad, act):
    if pad == 1:
        y = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="circular"), w, b)
    else:
        y = F.conv2d(x, w, b, padding=1)
    return F.leaky_relu(y) if act == 1 else (torch.sigmoid(y) if act == 2 else y)


def mask_words(y):
    """[B][ceil(C/32)][H][W] int32 words, bit j of word w = (channel 32w + j of y > 0)."""
    B, C, H, W = y.shape
    nw = (C + 31) // 32
    bits = (y > 0).to(torch.int64)
    if nw * 32 > C:
        bits = torch.cat([bits, torch.zeros(B, nw * 32 - C, H, W, device=y.device, dtype=torch.int64)], 1)
    words = (bits.view(B, nw, 32, H, W) << torch.arange(32, device=y.device).view(1, 1, 32, 1, 1)).sum(2)
    return torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32).contiguous()


@pytest.mark.parametrize("B,c1,c2,cout,H,W,pad,act", SHAPES)
def test_forward_and_input_gradient_vs_float64(K, B, c1, c2, cout, H, W, pad, act):
    torch.manual_seed(0)
    x1 = torch.randn(B, c1, H, W)
    x2 = torch.randn(B, c2, H, W) if c2 else None
    w = torch.randn(cout, c1 + c2, 3, 3) * 0.05
    b, sb, sc = torch.randn(cout), torch.randn(B, cout) * 0.3, torch.tensor([0.7])
    xin = x1 if x2 is None else torch.cat([x1, x2], 1)
    yr = ref_conv(xin.double(), w.double() * 0.7, b.double(), pad, 0) + sb.double()[:, :, None, None]
    yr = F.leaky_relu(yr) if act == 1 else (torch.sigmoid(yr) if act == 2 else yr)
    err, ys = {}, {}
    for wino in (1, 0):
        fp, _ = K.pack(w.cuda(), wino)
        mask = torch.zeros(B, (cout + 31) // 32, H, W, device="cuda", dtype=torch.int32) if (act == 1 and cout > 32) else None
        y1, y2 = K.conv(wino, x1.cuda(), None if x2 is None else x2.cuda(), fp, cout, pad, act, bias=b.cuda(), sbias=sb.cuda(),
                        scale=sc.cuda(), co1=128 if cout == 256 else None, mask_out=mask)
        y = y1 if y2 is None else torch.cat([y1, y2], 1)
        err[wino], ys[wino] = float((y.cpu().double() - yr).abs().max()), (y, mask)
    assert err[1] < 1e-5 and err[1] <= 2.0 * err[0] + 1e-7, err
    y, mask = ys[1]
    if mask is not None:            # the 1-bit gate is the sign of THIS kernel's output
        got, want = mask, mask_words(y)
        if cout % 32:
            keep = (1 << (cout % 32)) - 1
            got, want = got.clone(), want.clone()
            got[:, -1] &= keep
            want[:, -1] &= keep
        assert torch.equal(got, want)
    # input gradient: dx = conv(dy * lrelu'(y), dgrad pack), split into dx1 / dx2 with dx1 accumulated into a buffer;
    # float gate and 1-bit gate.  A gradient with <= 16 channels is the 16-row form's (conv3x3_wino16.hip): ungated,
    # one plain output -- what the layer code sends it (a gated one runs the direct kernel).
    thin = c1 + c2 <= 16
    xin_r = xin.double().requires_grad_(True)
    yr2 = ref_conv(xin_r, w.double(), None, pad, 0 if thin else 1)
    dy = torch.randn_like(yr2)
    yr2.backward(dy)
    yf = yr2.detach().float().cuda()
    for mode in (("none",) if thin else ("float", "mask")):
        if mode == "mask" and cout <= 32:
            continue
        e = {}
        for wino in (1, 0):
            _, dp = K.pack(w.cuda(), wino)
            base = torch.randn(B, c1, H, W, device="cuda") if (c2 and not thin) else None
            dx1, dx2 = K.conv(wino, dy.float().cuda(), None, dp, c1 + c2, pad, 0, gate1=None if thin else yf,
                              gate_act=0 if thin else 1, co1=c1 if (c2 and not thin) else None,
                              mask_in=mask_words(yf) if mode == "mask" else None, y1_acc=None if base is None else base.clone())
            if base is not None:
                dx1 = dx1 - base
            dx = dx1 if dx2 is None else torch.cat([dx1, dx2], 1)
            e[wino] = float((dx.cpu().double() - xin_r.grad).abs().max())
        assert e[1] < 2e-5 and e[1] <= 2.0 * e[0] + 1e-7, (mode, e)


THIN = [  # B, c1, c2, cout <= 16, H, W, circular, act: the 16-row form's forward
    (2, 128, 128, 16, 16, 16, 1, 2), (2, 128, 0, 16, 16, 16, 0, 2), (1, 40, 0, 16, 20, 12, 1, 0), (2, 64, 0, 3, 9, 8, 0, 1),
    (1, 128, 128, 16, 64, 64, 1, 2), (1, 32, 0, 16, 128, 128, 0, 0), (1, 32, 8, 12, 33, 22, 1, 2), (3, 128, 0, 16, 32, 32, 1, 0),
    (2, 128, 128, 16, 84, 84, 1, 2), (2, 128, 0, 16, 84, 84, 0, 2), (1, 40, 0, 5, 84, 84, 1, 1)]


@pytest.mark.parametrize("B,c1,c2,cout,H,W,pad,act", THIN)
def test_sixteen_channel_form_vs_float64(K, B, c1, c2, cout, H, W, pad, act):
    """conv3x3_wino16.hip (v_mfma_f32_16x16x4_f32; Transition.conv6, Encoder.conv4): forward with scale, bias and every
    activation against float64, next to the direct 16-row kernel."""
    assert K.lib.scm_conv3x3_wino16_serves(H, W) == 1
    torch.manual_seed(2)
    x1 = torch.randn(B, c1, H, W)
    x2 = torch.randn(B, c2, H, W) if c2 else None
    w, b, sc = torch.randn(cout, c1 + c2, 3, 3) * 0.05, torch.randn(cout), torch.tensor([0.7])
    xin = x1 if x2 is None else torch.cat([x1, x2], 1)
    yr = ref_conv(xin.double(), w.double() * 0.7, b.double(), pad, act)
    err = {}
    for wino in (1, 0):
        fp, _ = K.pack(w.cuda(), wino)
        y, _ = K.conv(wino, x1.cuda(), None if x2 is None else x2.cuda(), fp, cout, pad, act, bias=b.cuda(), scale=sc.cuda())
        err[wino] = float((y.cpu().double() - yr).abs().max())
    assert err[1] < 1e-5 and err[1] <= 2.0 * err[0] + 1e-7, err
    assert K.lib.scm_conv3x3_wino16_serves(84, 84) == 1          # (three pixel pairs per thread: BASELINE configs[3] frames)


@pytest.mark.parametrize("B,c1,c2,cout,H,W,pad,act", WSHAPES)
def test_weight_gradient_vs_float64(K, B, c1, c2, cout, H, W, pad, act):
    torch.manual_seed(1)
    x1 = torch.randn(B, c1, H, W)
    x2 = torch.randn(B, c2, H, W) if c2 else None
    w = (torch.randn(cout, c1 + c2, 3, 3) * 0.05).double().requires_grad_(True)
    bb = torch.randn(cout).double().requires_grad_(True)
    xin = (x1 if x2 is None else torch.cat([x1, x2], 1)).double()
    y = ref_conv(xin, w, bb, pad, act)
    dy = torch.randn_like(y)
    y.backward(dy)
    yf = y.detach().float().cuda()
    sw, sb = max(1.0, float(w.grad.abs().max())), max(1.0, float(bb.grad.abs().max()))
    for mode in (("float", "mask") if (act == 1 and cout > 32) else ("float",)):
        e = {}
        for wino in (1, 0):
            dw, db, csum = K.wgrad(wino, x1.cuda(), None if x2 is None else x2.cuda(), dy.float().cuda(), yf, act, pad,
                                   mask=mask_words(yf) if mode == "mask" else None)
            e[wino] = (float((dw.cpu().double() - w.grad).abs().max()) / sw, float((db.cpu().double() - bb.grad).abs().max()) / sb,
                       float((csum.sum(0).cpu().double() - bb.grad).abs().max()) / sb)
        assert max(e[1]) < 2e-6 and e[1][0] <= 3.0 * e[0][0] + 1e-8, (mode, e)


KTHIN = [  # B, cin <= 16, cout, H, W, circular, act, gate_act (input gradient form), co1 (split), accumulate
    (2, 16, 128, 16, 16, 1, 1, 0, None, False), (2, 16, 256, 64, 64, 1, 0, 2, 128, False), (1, 9, 72, 20, 12, 0, 1, 0, None, False),
    (3, 12, 100, 9, 8, 1, 0, 1, 64, True), (1, 16, 64, 33, 22, 0, 1, 0, None, False), (2, 16, 48, 6, 6, 1, 0, 2, 16, False),
    (1, 16, 128, 128, 128, 1, 1, 0, None, False)]


@pytest.mark.parametrize("B,cin,cout,H,W,pad,act,gact,co1,acc", KTHIN)
def test_sixteen_input_channel_form_vs_float64(K, B, cin, cout, H, W, pad, act, gact, co1, acc):
    """conv3x3_wino_k16.hip (Transition.conv1's forward, conv6's input gradient): scale, bias, per-sample bias, leaky-relu
    + the 1-bit gate words it writes, float-gated inputs (leaky-relu / sigmoid), split and accumulated outputs, ragged
    channel counts, odd heights, both paddings -- against float64 next to the wide Winograd kernel (SCM_NO_WINO_K16 is read
    once per process, so the comparison partner here is the direct kernel)."""
    torch.manual_seed(6)
    x = torch.randn(B, cin, H, W)
    w, b, sb, sc = torch.randn(cout, cin, 3, 3) * 0.05, torch.randn(cout), torch.randn(B, cout) * 0.3, torch.tensor([0.8])
    gate = torch.rand(B, cin, H, W) - (0.5 if gact == 1 else 0.0) if gact else None
    xin = x.double()
    if gact == 1:
        xin = xin * torch.where(gate > 0, 1.0, 0.01).double()
    if gact == 2:
        xin = xin * (gate * (1 - gate)).double()
    yr = ref_conv(xin, w.double() * 0.8, None, pad, 0) + (b.double()[None, :, None, None] + sb.double()[:, :, None, None])
    yr = F.leaky_relu(yr) if act == 1 else yr
    base = torch.randn(B, co1 or cout, H, W) if acc else None
    err, masks = {}, {}
    for wino in (1, 0):
        fp, _ = K.pack(w.cuda(), wino)
        mo = torch.zeros(B, (cout + 31) // 32, H, W, device="cuda", dtype=torch.int32) if (act == 1 and cout > 32) else None
        y1, y2 = K.conv(wino, x.cuda(), None, fp, cout, pad, act, bias=b.cuda(), sbias=sb.cuda(), scale=sc.cuda(),
                        gate1=None if gate is None else gate.cuda(), gate_act=gact, co1=co1, mask_out=mo,
                        y1_acc=None if base is None else base.cuda().clone())
        y = y1 if y2 is None else torch.cat([y1, y2], 1)
        ref = yr.clone()
        if acc:
            ref[:, :base.shape[1]] += base.double()
        err[wino] = float((y.cpu().double() - ref).abs().max())
        if mo is not None:
            assert torch.equal(mo, mask_words(y))
    assert err[1] < 2e-5 and err[1] <= 2.0 * err[0] + 1e-6, err


WTHIN = [  # B, c1, c2, cout, H, W, circular, act: at most 16 channels on one side (conv3x3_wgrad_wino16.hip), and shapes
    # it hands to the direct kernel (W % 32 != 0)
    (2, 128, 128, 16, 64, 64, 1, 2), (2, 16, 0, 128, 64, 64, 1, 0), (2, 128, 0, 16, 32, 32, 0, 2), (1, 16, 0, 128, 32, 64, 0, 1),
    (3, 40, 0, 9, 8, 32, 1, 0), (2, 9, 0, 70, 6, 32, 1, 1), (1, 64, 32, 12, 16, 128, 1, 1), (2, 16, 0, 16, 34, 32, 0, 0),
    (1, 128, 0, 16, 128, 128, 1, 2), (5, 16, 0, 128, 64, 64, 1, 0), (2, 128, 0, 16, 20, 12, 1, 0), (1, 16, 0, 64, 84, 84, 1, 1),
    # ragged last 32-column part (W % 4 == 0, W > 32): 84 x 84 (BASELINE configs[3]) in both directions, both paddings, gates
    (2, 128, 128, 16, 84, 84, 1, 2), (2, 16, 0, 128, 84, 84, 1, 0), (2, 128, 0, 16, 84, 84, 0, 2), (1, 9, 0, 128, 84, 84, 0, 1),
    (2, 64, 0, 16, 10, 36, 1, 1), (1, 16, 0, 96, 12, 60, 0, 0)]


@pytest.mark.parametrize("B,c1,c2,cout,H,W,pad,act", WTHIN)
def test_thin_weight_gradient_vs_float64(K, B, c1, c2, cout, H, W, pad, act):
    """Transition.conv6 / conv1, Encoder.conv4 shaped weight gradients (16 channels on one side): the Winograd form on
    v_mfma_f32_16x16x4_f32 against float64 next to the direct kernel -- dw, db, per-sample channel sums; ungated, float
    leaky-relu and sigmoid gates, both paddings, ragged channel counts, two input tensors, odd patch-row counts; twice,
    bitwise equal (fixed summation order)."""
    torch.manual_seed(4)
    x1 = torch.randn(B, c1, H, W)
    x2 = torch.randn(B, c2, H, W) if c2 else None
    w = (torch.randn(cout, c1 + c2, 3, 3) * 0.05).double().requires_grad_(True)
    bb = torch.randn(cout).double().requires_grad_(True)
    xin = (x1 if x2 is None else torch.cat([x1, x2], 1)).double()
    y = ref_conv(xin, w, bb, pad, act)
    dy = torch.randn_like(y)
    y.backward(dy)
    yf = y.detach().float().cuda()
    sw, sb = max(1.0, float(w.grad.abs().max())), max(1.0, float(bb.grad.abs().max()))
    e, first = {}, None
    for wino in (1, 1, 0):
        dw, db, csum = K.wgrad(wino, x1.cuda(), None if x2 is None else x2.cuda(), dy.float().cuda(), yf if act else None, act, pad)
        if wino and first is None:
            first = (dw.clone(), db.clone(), csum.clone())
        elif wino:
            assert torch.equal(dw, first[0]) and torch.equal(db, first[1]) and torch.equal(csum, first[2])
        e[wino] = (float((dw.cpu().double() - w.grad).abs().max()) / sw, float((db.cpu().double() - bb.grad).abs().max()) / sb,
                   float((csum.sum(0).cpu().double() - bb.grad).abs().max()) / sb)
    assert max(e[1]) < 2e-6 and e[1][0] <= 3.0 * e[0][0] + 1e-8, e


def test_bitwise_reproducible_and_layer_fallback():
    """Two launches give identical bits (fixed summation order, no atomics); a layer on a frame the kernels do not take
    (odd width) runs the direct kernels through the same autograd function and agrees with the direct algorithm bitwise."""
    from scm_gan_amd import ops
    from scm_gan_amd.layers import Conv2d
    assert ops.conv_algo() in ("winograd", "direct")
    prev = ops.conv_algo()
    try:
        outs = {}
        for algo in ("winograd", "winograd", "direct"):
            ops.set_conv_algo(algo)
            torch.manual_seed(3)
            layer = Conv2d(128, 128, 3, padding=2, padding_mode="circular").cuda()
            res = []
            for (H, W) in ((16, 16), (9, 7)):
                x = torch.randn(2, 128, H, W, device="cuda", requires_grad=True)
                yy = layer(x, act=ops.ACT_LRELU)
                yy.backward(torch.ones_like(yy) * 1e-3)
                res += [yy.detach().clone(), x.grad.clone(), layer.weight.grad.clone()]
                layer.weight.grad = None
            outs.setdefault(algo, []).append(res)
        a, b = outs["winograd"]
        assert all(torch.equal(p, q) for p, q in zip(a, b))
        d = outs["direct"][0]
        # 16 x 16: two algorithms, fp32-rounding apart; 9 x 7: both ran the direct kernels
        assert all(float((p - q).abs().max()) < 1e-4 * max(1.0, float(q.abs().max())) for p, q in zip(a[:3], d[:3]))
        assert not torch.equal(a[0], d[0])
        assert all(torch.equal(p, q) for p, q in zip(a[3:], d[3:]))
    finally:
        ops.set_conv_algo(prev)


def test_output_gate_of_the_input_gradient_kernel(K):
    """gate_mask_y1 / _y2: what the Winograd kernel writes (or adds) to y1 / y2 is multiplied by the leaky-relu slope of
    the element it writes to -- bitwise the ungated result times 1 or 0.01, for split outputs, with accumulate, on
    the fast and the generic epilogue (ragged channels, odd height)."""
    from scm_gan_amd import _lib as L
    torch.manual_seed(4)
    for (B, cin, c1, c2, H, W, acc) in [(2, 128, 128, 128, 16, 16, True), (1, 16, 128, 128, 64, 64, False),
                                        (2, 64, 128, 0, 16, 16, False), (2, 64, 96, 64, 9, 8, True)]:
        cout = c1 + c2
        dy = torch.randn(B, cin, H, W, device="cuda")
        w = torch.randn(cin, cout, 3, 3, device="cuda") * 0.05          # the layer's weight: cout inputs -> cin outputs
        _, dp = K.pack(w, 1)
        a1 = torch.randn(B, c1, H, W, device="cuda")                     # activations whose gradients y1 / y2 are
        a2 = torch.randn(B, c2, H, W, device="cuda") if c2 else None
        m1, m2 = mask_words(a1), (mask_words(a2) if c2 else None)

        def run(masks):
            base = torch.randn(B, c1, H, W, device="cuda", generator=torch.Generator("cuda").manual_seed(1)) if acc else None
            y1 = base.clone() if acc else torch.empty(B, c1, H, W, device="cuda")
            y2 = torch.empty(B, c2, H, W, device="cuda") if c2 else None
            d = L.ConvDesc(K.P(dy), None, cin, 0, None, None, 0, K.P(dp), cout, None, None, None, K.P(y1), K.P(y2), c1, B, H, W,
                           1, 0, None, None, 1 if acc else 0, K.P(m1) if masks else None, K.P(m2) if (masks and c2) else None)
            L.check(K.lib.scm_conv3x3_wino(ctypes.byref(d), K.S()), "conv")
            return (y1 - base if acc else y1), y2, base
        g1, g2, _ = run(True)
        u1, u2, _ = run(False)
        s1 = torch.where(a1 > 0, torch.ones_like(a1), torch.full_like(a1, 0.01))
        if acc:         # (y1 = base + gated: compare after removing base, to fp32 rounding of the add)
            assert float((g1 - u1 * s1).abs().max()) < 1e-6 * max(1.0, float(u1.abs().max()))
        else:
            assert torch.equal(g1, u1 * s1)
        if c2:
            s2 = torch.where(a2 > 0, torch.ones_like(a2), torch.full_like(a2, 0.01))
            assert torch.equal(g2, u2 * s2)
    # the direct entry point refuses output gates instead of ignoring them
    d = L.ConvDesc(K.P(dy), None, cin, 0, None, None, 0, K.P(dp), cout, None, None, None, K.P(a1), K.P(a2), c1, B, H, W, 1, 0,
                   None, None, 0, K.P(m1), None)
    assert K.lib.scm_conv3x3(ctypes.byref(d), K.S()) != 0


def test_producer_side_gating_equals_consumer_side():
    """Encoder + Transition backward with the gate applied by the producers of each gradient (default) against the same
    pass with every layer gating its own incoming gradient (ops.use_pregate = False): same gradients to fp32 rounding
    (a two-consumer sum is gated per contribution instead of once)."""
    import weights as WS
    from scm_gan_amd import models as M, ops
    if ops.conv_algo() != "winograd" or ops.compute_precision() != "fp32":
        pytest.skip("producer-side gating belongs to the Winograd kernels (SCM_CONV_ALGO / SCM_PRECISION select another path)")
    torch.manual_seed(0)
    res = {}
    for pre in (True, False):
        ops.use_pregate = pre
        try:
            enc, tr = M.Encoder(16, 3), M.Transition(16, 5)
            enc.load_state_dict(WS.to_torch(WS.encoder_state(30)))
            tr.load_state_dict(WS.to_torch(WS.transition_state(31)))
            enc.train(); tr.train()
            g = torch.Generator("cuda").manual_seed(5)
            x = (torch.rand(2, 3, 3, 32, 32, device="cuda", generator=g) < 0.3).float()
            a = torch.eye(5, device="cuda")[[1, 3]]
            tr.set_noise([torch.rand(2, 16, 32, 32, device="cuda", generator=g)])
            z = enc(x)
            assert ops.pregate_ok(32, 32) == pre
            z2 = tr(z, a)
            (z2 * torch.rand(z2.shape, device="cuda", generator=g)).sum().backward()
            res[pre] = [p.grad.clone() for m in (enc, tr) for p in m.parameters() if p.grad is not None]
        finally:
            ops.use_pregate = True
    assert len(res[True]) == len(res[False]) >= 20
    for p, q in zip(res[True], res[False]):
        assert float((p - q).abs().max()) <= 2e-6 * max(1e-6, float(q.abs().max())), float((p - q).abs().max())


@pytest.mark.parametrize("B,c1,c2,cout,H,pad,mode", [
    (32, 128, 0, 128, 64, 1, "forward"), (32, 128, 128, 128, 64, 1, "forward"), (16, 128, 0, 256, 64, 0, "forward"),
    (256, 64, 0, 128, 8, 1, "forward"), (16, 128, 0, 256, 64, 1, "gated_dgrad"), (256, 128, 0, 128, 12, 0, "gated_dgrad_acc")])
def test_two_tiles_per_workgroup_is_bitwise_the_one_tile_kernel(K, B, c1, c2, cout, H, pad, mode, monkeypatch):
    """conv3x3_wino_t2.hip (64-wide frames, H % 4 == 0, 128-channel groups, ungated input: every wide launch of the
    transition / encoder at 64 x 64): a workgroup takes two vertically adjacent tiles, the second tile's first rows
    requested before the first tile's epilogue.  Same pack, same order of operations per output: bitwise the results of
    conv3x3_wino_kernel<1, 0, PADC, 128> (SCM_WINO_T2=0) -- forward with bias / scale / per-sample bias and the 1-bit gate
    it writes, two input tensors, split outputs; input gradients with the destination's gate words, with accumulate;
    wrapped (circular) and zero-padded first / last tiles."""
    from scm_gan_amd import _lib as L
    W = 64
    wg2 = B * (H // 4) * (cout // 128)         # (scm_wino_t2_serves' conditions: the launch takes the two-tile kernel)
    assert H % 4 == 0 and wg2 >= 512 and -(-wg2 // 256) * 1644 <= -(-2 * wg2 // 256) * 838
    torch.manual_seed(H + cout)
    x1 = torch.randn(B, c1, H, W, device="cuda")
    x2 = torch.randn(B, c2, H, W, device="cuda") if c2 else None
    w = torch.randn(cout, c1 + c2, 3, 3, device="cuda") * 0.05
    fp, _ = K.pack(w, 1)
    bias = torch.randn(cout, device="cuda")
    sbias = torch.randn(B, cout, device="cuda")
    scale = torch.full((1,), 0.7, device="cuda")
    co1 = 128 if cout == 256 else cout
    a1, a2 = torch.randn(B, co1, H, W, device="cuda"), (torch.randn(B, cout - co1, H, W, device="cuda") if co1 < cout else None)
    m1, m2 = mask_words(a1), (mask_words(a2) if a2 is not None else None)

    def run(t2):
        monkeypatch.setenv("SCM_WINO_T2", "1" if t2 else "0")
        if mode == "forward":
            mask = torch.zeros(B, cout // 32, H, W, device="cuda", dtype=torch.int32)
            y1, y2 = K.conv(1, x1, x2, fp, cout, pad, 1, bias=bias, sbias=sbias, scale=scale, co1=co1, mask_out=mask)
            return [y1, y2, mask]
        acc = mode.endswith("acc")
        y1 = torch.full((B, co1, H, W), 0.25, device="cuda") if acc else torch.empty(B, co1, H, W, device="cuda")
        y2 = torch.empty(B, cout - co1, H, W, device="cuda") if co1 < cout else None
        d = L.ConvDesc(K.P(x1), None, c1, 0, None, None, 0, K.P(fp), cout, K.P(scale), None, None, K.P(y1), K.P(y2), co1, B, H, W,
                       pad, 0, None, None, 1 if acc else 0, K.P(m1), K.P(m2))
        L.check(K.lib.scm_conv3x3_wino(ctypes.byref(d), K.S()), "conv")
        return [y1, y2]
    two, one = run(True), run(False)
    for t, o in zip(two, one):
        assert (t is None) == (o is None)
        if t is not None:
            assert torch.equal(t, o)
    # (and the float64 reference, so that the pair is not only self-consistent)
    if mode == "forward":
        xx = x1 if x2 is None else torch.cat([x1, x2], 1)
        ref = ref_conv(xx.double(), w.double() * 0.7, None, pad, 0) + (bias.double().view(1, -1, 1, 1) + sbias.double().view(B, -1, 1, 1))
        ref = F.leaky_relu(ref)
        got = two[0] if two[1] is None else torch.cat([two[0], two[1]], 1)
        assert float((got.double() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
