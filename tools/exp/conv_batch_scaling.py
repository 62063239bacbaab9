"""Where the fixed cost of a conv launch sits: time vs batch (workgroup rounds) at 128 -> 128 and 16 -> 128."""
import sys, torch
sys.path.insert(0, '.')
from scm_gan_amd import ops
H = W = 64
def t(fn, n=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for cin in (128, 16):
    cout = 128
    w = torch.randn(cout, cin, 3, 3, device='cuda') * 0.05
    bias = torch.zeros(cout, device='cuda')
    fp, dp, fprec, dprec = ops.pack_conv3x3(w, cin, cout)
    for B in (8, 16, 32, 64, 96, 128, 256):
        x = torch.randn(B, cin, H, W, device='cuda')
        us = t(lambda: ops._conv3x3_raw(x, None, None, None, 0, fp, cout, None, bias, None, cout, 1, ops.ACT_LRELU))
        wgs = B * 16
        print("cin %3d B %3d: %7.1f us  workgroups %5d (%.2f rounds of 512)  us per round %.1f" % (cin, B, us, wgs, wgs / 512, us / (wgs / 512)))
