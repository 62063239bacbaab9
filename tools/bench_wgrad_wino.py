"""Timing of the Winograd weight-gradient entry point (kernel + fixed-order reduce) on the step's shapes.
SCM_NO_WGRAD_WINOZ=1 python tools/bench_wgrad_wino.py   -> the first form (conv3x3_wgrad_wino.hip) for an A/B."""
import ctypes, os, sys, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tools')
import bench_kernels as BK
from scm_gan_amd import _lib as L
lib, P, S = BK.lib, BK.P, BK.S


def wg(B, c1, c2, cout, H=64, W=64, gated=0, pad=1, n=30):
    cin = c1 + c2
    x1 = torch.randn(B, c1, H, W, device='cuda'); x2 = torch.randn(B, c2, H, W, device='cuda') if c2 else None
    dy = torch.randn(B, cout, H, W, device='cuda'); y = torch.randn(B, cout, H, W, device='cuda')
    pf, cf = ctypes.c_size_t(), ctypes.c_size_t()
    lib.scm_conv3x3_wgrad_workspace(B, H, W, cin, cout, ctypes.byref(pf), ctypes.byref(cf))
    part = torch.empty(pf.value, device='cuda'); cws = torch.empty(cf.value, device='cuda')
    dw = torch.empty(cout, cin, 3, 3, device='cuda'); db = torch.empty(cout, device='cuda')
    d3 = L.WgradDesc(P(x1), P(x2), c1, c2, P(dy), P(y) if gated else None, gated, cout, P(dw), P(db), None, P(part), P(cws),
                     B, H, W, pad, 0, 0, 0, 0, None, None)
    t = BK.timeit(lambda: L.check(lib.scm_conv3x3_wgrad_wino(ctypes.byref(d3), S()), 'wgrad'), n)
    fl = 2.0 * B * H * W * cout * cin * 9
    print('wgrad_wino B %3d %3d+%3d -> %3d %3dx%-3d gate %d pad %d: %7.1f us (kernel + reduce)  %6.1f TF algorithmic'
          % (B, c1, c2, cout, H, W, gated, pad, t, fl / t / 1e6))


if __name__ == '__main__':
    print('form:', 'first (SCM_NO_WGRAD_WINOZ)' if os.environ.get('SCM_NO_WGRAD_WINOZ') else 'zigzag')
    wg(64, 128, 0, 128); wg(64, 128, 128, 128); wg(64, 128, 0, 128, pad=0); wg(64, 128, 0, 128, gated=1)
    wg(32, 128, 0, 128, 128, 128); wg(8, 128, 0, 128)
