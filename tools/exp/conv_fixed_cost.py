"""Fixed cost of the fp32 MFMA conv launch: time vs input channels at cout = 128 / 256 / 16 (B=64, 64x64)."""
import sys, torch
sys.path.insert(0, '.')
from scm_gan_amd import ops
B, H, W = 64, 64, 64
def t(fn, n=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for cout in (128, 256, 16):
    for cin in (16, 32, 64, 128, 256):
        x = torch.randn(B, cin, H, W, device='cuda'); w = torch.randn(cout, cin, 3, 3, device='cuda') * 0.05
        bias = torch.zeros(cout, device='cuda')
        fp, dp, fprec, dprec = ops.pack_conv3x3(w, cin, cout)
        us = t(lambda: ops._conv3x3_raw(x, None, None, None, 0, fp, cout, None, bias, None, cout, 1, ops.ACT_LRELU))
        flops = 2.0 * B * H * W * cout * cin * 9
        print("cin %3d -> cout %3d: %7.1f us  %6.1f TFLOP/s  (ideal at 157 TF: %6.1f us)" % (cin, cout, us, flops / us / 1e6, flops / 157.3e6))
