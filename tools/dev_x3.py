"""Dev check of the bf16x3-split fp32 convolution: error vs float64 next to the native fp32 kernel, and time."""
import ctypes, sys, torch
sys.path.insert(0, '.')
from scm_gan_amd import _lib as L, ops
lib = L.lib()
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def packs_x3(w, cin, cout):
    fp = torch.empty(lib.scm_conv3x3_pack_x3_bytes(cin, cout) // 2, device='cuda', dtype=torch.int16)
    dp = torch.empty(lib.scm_conv3x3_pack_x3_bytes(cout, cin) // 2, device='cuda', dtype=torch.int16)
    L.check(lib.scm_pack_conv3x3_weights_x3(P(w), cin, cout, 0, 0, 0, P(fp), P(dp), S()), 'pack_x3')
    return fp, dp
def conv_x3(x, pack, cout, bias, pad, act, gate=None, gate_act=0, mask_in=None, mask_out=None):
    B, C, H, W = x.shape
    y = torch.empty(B, cout, H, W, device='cuda')
    d = L.ConvDesc(P(x), None, C, 0, P(gate), None, gate_act, P(pack), cout, None, P(bias), None, P(y), None, cout, B, H, W, pad, act, P(mask_out), P(mask_in), 0)
    L.check(lib.scm_conv3x3_x3(ctypes.byref(d), S()), 'conv_x3')
    return y
def t(fn, n=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
torch.manual_seed(0)
import torch.nn.functional as F
for (B, cin, cout, H, W, circ) in [(2, 16, 128, 16, 16, 1), (1, 128, 128, 20, 12, 1), (2, 256, 128, 16, 16, 1), (1, 9, 128, 33, 20, 0), (2, 128, 16, 16, 16, 0), (1, 128, 256, 7, 13, 1)]:
    x = (torch.rand(B, cin, H, W, device='cuda') < 0.5).float() * torch.randn(B, cin, H, W, device='cuda')
    w = torch.randn(cout, cin, 3, 3, device='cuda') * (1.0 / (cin * 9) ** 0.5)
    bias = torch.randn(cout, device='cuda') * 0.1
    fp32p, dp32, _, _ = ops.pack_conv3x3(w, cin, cout)
    y_nat, _ = ops._conv3x3_raw(x, None, None, None, 0, fp32p, cout, None, bias, None, cout, circ, ops.ACT_LRELU)
    fx, dx = packs_x3(w, cin, cout)
    y_x3 = conv_x3(x, fx, cout, bias, circ, ops.ACT_LRELU)
    xd, wd = x.double().cpu(), w.double().cpu()
    xp = F.pad(xd, (1, 1, 1, 1), mode='circular') if circ else F.pad(xd, (1, 1, 1, 1))
    ref = F.leaky_relu(F.conv2d(xp, wd, bias.double().cpu()), 0.01)
    en = float((y_nat.double().cpu() - ref).abs().max()); ex = float((y_x3.double().cpu() - ref).abs().max())
    print("B=%d %d->%d %dx%d circ=%d: max err vs float64: native f32 %.2e, bf16x3 %.2e  (|y| max %.2f)" % (B, cin, cout, H, W, circ, en, ex, float(ref.abs().max())))
    assert ex < 3 * en + 1e-6
B, C, H, W = 64, 128, 64, 64
x = torch.randn(B, C, H, W, device='cuda'); w = torch.randn(C, C, 3, 3, device='cuda') * 0.03; bias = torch.zeros(C, device='cuda')
fp32p, _, _, _ = ops.pack_conv3x3(w, C, C)
fx, dx = packs_x3(w, C, C)
print("128->128 B=64 64x64: native f32 %.1f us, bf16x3 %.1f us" % (
    t(lambda: ops._conv3x3_raw(x, None, None, None, 0, fp32p, C, None, bias, None, C, 1, ops.ACT_LRELU)),
    t(lambda: conv_x3(x, fx, C, bias, 1, ops.ACT_LRELU))))
