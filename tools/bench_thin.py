"""Timing of the 16-channel-side layers of the transition (conv6 256 -> 16, conv1 16 -> 128) through the Winograd entry
points: forward, input gradient, weight gradient (kernel + reduce)."""
import ctypes, sys, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
import bench_kernels as BK
from scm_gan_amd import _lib as L, ops
lib, P, S = BK.lib, BK.P, BK.S
B, H, W, C = 64, 64, 64, 128
torch.manual_seed(0)
xa, xb = torch.randn(B, C, H, W, device='cuda'), torch.randn(B, C, H, W, device='cuda')
w6 = torch.randn(16, 2 * C, 3, 3, device='cuda') * 0.05
w1 = torch.randn(C, 16, 3, 3, device='cuda') * 0.05
z16 = torch.randn(B, 16, H, W, device='cuda'); g16 = torch.randn(B, 16, H, W, device='cuda') * 1e-4
gy = torch.randn(B, C, H, W, device='cuda') * 1e-4
p6, p1 = ops.pack_conv3x3(w6, 2 * C, 16), ops.pack_conv3x3(w1, 16, C)
mask = torch.zeros(B, C // 32, H, W, device='cuda', dtype=torch.int32)
b16, b128 = torch.zeros(16, device='cuda'), torch.zeros(C, device='cuda')
pf, cf = ctypes.c_size_t(), ctypes.c_size_t()
lib.scm_conv3x3_wgrad_workspace(B, H, W, 2 * C, 16, ctypes.byref(pf), ctypes.byref(cf))
part = torch.empty(pf.value * 2, device='cuda'); cws = torch.empty(cf.value * 2 + 65536, device='cuda')
dw6 = torch.empty_like(w6); db6 = torch.empty(16, device='cuda'); dw1 = torch.empty_like(w1); db1 = torch.empty(C, device='cuda')
prob, _ = ops._conv3x3_raw(xa, xb, None, None, 0, p6[0], 16, None, b16, None, 16, 1, ops.ACT_SIGMOID, p6[2])
d6 = L.WgradDesc(P(xa), P(xb), C, C, P(g16), P(prob), ops.ACT_SIGMOID, 16, P(dw6), P(db6), None, P(part), P(cws), B, H, W, 1, 0, 0, 0, 0, None, None, None)
d1 = L.WgradDesc(P(z16), None, 16, 0, P(gy), None, 0, C, P(dw1), P(db1), None, P(part), P(cws), B, H, W, 1, 0, 0, 0, 0, None, None, None)
cases = [
    ('conv6 forward 256 -> 16 (sigmoid)', lambda: ops._conv3x3_raw(xa, xb, None, None, 0, p6[0], 16, None, b16, None, 16, 1, ops.ACT_SIGMOID, p6[2])),
    ('conv6 input gradient 16 -> 256 (sigmoid gate, two masks)', lambda: ops._conv3x3_raw(g16, None, prob, None, ops.ACT_SIGMOID, p6[1], 2 * C, None, None, None, C, 1, ops.ACT_NONE, p6[3], ymask1=mask, ymask2=mask)),
    ('conv6 weight gradient', lambda: L.check(lib.scm_conv3x3_wgrad_wino(ctypes.byref(d6), S()), 'w6')),
    ('conv1 forward 16 -> 128 (leaky-relu + mask)', lambda: ops._conv3x3_raw(z16, None, None, None, 0, p1[0], C, None, b128, None, C, 1, ops.ACT_LRELU, p1[2], mask_out=mask)),
    ('conv1 input gradient 128 -> 16', lambda: ops._conv3x3_raw(gy, None, None, None, 0, p1[1], 16, None, None, None, 16, 1, ops.ACT_NONE, p1[3])),
    ('conv1 weight gradient', lambda: L.check(lib.scm_conv3x3_wgrad_wino(ctypes.byref(d1), S()), 'w1')),
]
for name, fn in cases:
    print('%-62s %7.1f us' % (name, BK.timeit(fn, 30)))
