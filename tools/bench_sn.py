"""Timing of scm_sn_power_iter on the transition's five layers (one workgroup per layer) for 1 and 7 back-to-back calls."""
import sys, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
import bench_kernels as BK
from scm_gan_amd import ops
torch.manual_seed(0)
shapes = [(128, 21 * 9), (128, 1152), (128, 1152), (128, 1152), (128, 2304)]
layers = []
for r, c in shapes:
    w = torch.randn(r, c, device='cuda') * 0.05
    u = torch.nn.functional.normalize(torch.randn(r, device='cuda'), dim=0)
    v = torch.nn.functional.normalize(torch.randn(c, device='cuda'), dim=0)
    layers.append((w, u, v))
for steps in (1, 7):
    t = BK.timeit(lambda: ops.sn_power_iter_multi(layers, steps), 20)
    print('sn_power_iter: 5 transition layers x %d call(s): %.1f us' % (steps, t))
enc = [(torch.randn(128, 81, device='cuda') * 0.05, torch.randn(128, device='cuda'), torch.randn(81, device='cuda')),
       layers[1], layers[2]]
t = BK.timeit(lambda: ops.sn_power_iter_multi(enc, 1), 20)
print('sn_power_iter: 3 encoder layers x 1 call: %.1f us' % t)
# float64 check of one call on the 128 x 1152 layer
w, u, v = layers[1]
u0, v0 = u.clone(), v.clone()
out = ops.sn_power_iter_multi([(w, u, v)], 1)
wd, ud = w.double(), u0.double()
vn = wd.t() @ ud; vn = vn / (vn.norm() + 1e-12)
s = wd @ vn; un = s / (s.norm() + 1e-12); sigma = float(un @ s)
print('check: |v - v64| %.2e |u - u64| %.2e sigma %.7f vs %.7f' % (float((v.double() - vn).abs().max()), float((u.double() - un).abs().max()),
                                                                   float(out[0, 0, 0]), sigma))
