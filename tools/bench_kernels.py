"""Stable per-kernel timings: direct C-ABI calls on preallocated buffers, HIP events."""
import ctypes, sys, torch
sys.path.insert(0, '.')
from scm_gan_amd import _lib as L, ops
lib = L.lib()
dev = 'cuda'
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

def timeit(fn, n=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

def case(B, c1, c2, cout, H=64, W=64, which=('fwd', 'dgrad', 'wgrad'), n=10):
    cin = c1 + c2
    x1 = torch.randn(B, c1, H, W, device=dev); x2 = torch.randn(B, c2, H, W, device=dev) if c2 else None
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05; b = torch.zeros(cout, device=dev)
    fp, dp = ops.pack_conv3x3(w, cin, cout)[:2]
    y = torch.empty(B, cout, H, W, device=dev); dy = torch.randn(B, cout, H, W, device=dev)
    dx1 = torch.empty(B, c1, H, W, device=dev); dx2 = torch.empty(B, c2, H, W, device=dev) if c2 else None
    fl = 2.0 * B * H * W * cout * cin * 9
    out = []
    if 'fwd' in which:
        d = L.ConvDesc(P(x1), P(x2), c1, c2, None, None, 0, P(fp), cout, None, P(b), None, P(y), None, cout, B, H, W, 1, 1)
        t = timeit(lambda: L.check(lib.scm_conv3x3(ctypes.byref(d), S()), 'conv'), n); out.append('fwd %7.1f us %6.1f TF' % (t, fl / t / 1e6))
    if 'dgrad' in which:
        d2 = L.ConvDesc(P(dy), None, cout, 0, P(y), None, 1, P(dp), cin, None, None, None, P(dx1), P(dx2), c1, B, H, W, 1, 0)
        t = timeit(lambda: L.check(lib.scm_conv3x3(ctypes.byref(d2), S()), 'dgrad'), n); out.append('dgrad %7.1f us %6.1f TF' % (t, fl / t / 1e6))
    if 'wgrad' in which:
        pf, cf = ctypes.c_size_t(), ctypes.c_size_t()
        lib.scm_conv3x3_wgrad_workspace(B, H, W, cin, cout, ctypes.byref(pf), ctypes.byref(cf))
        part = torch.empty(pf.value, device=dev); cws = torch.empty(cf.value, device=dev)
        dw = torch.empty_like(w); db = torch.empty(cout, device=dev)
        d3 = L.WgradDesc(P(x1), P(x2), c1, c2, P(dy), P(y), 1, cout, P(dw), P(db), None, P(part), P(cws), B, H, W, 1, 0, 0, 0, 0, None, None)
        t = timeit(lambda: L.check(lib.scm_conv3x3_wgrad(ctypes.byref(d3), S()), 'wgrad'), n); out.append('wgrad %7.1f us %6.1f TF' % (t, fl / t / 1e6))
    print('B %3d %3d+%3d -> %3d  %dx%d | ' % (B, c1, c2, cout, H, W) + ' | '.join(out)); sys.stdout.flush()

if __name__ == '__main__':
    which = tuple(sys.argv[1].split(',')) if len(sys.argv) > 1 else ('fwd', 'dgrad', 'wgrad')
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    case(64, 128, 0, 128, which=which, n=n)
    case(64, 128, 128, 128, which=which, n=n)
    case(64, 128, 128, 16, which=which, n=n)
    case(64, 16, 0, 128, which=which, n=n)
    case(64, 16, 0, 64, which=which, n=n)
    case(64, 64, 0, 3, which=which, n=n)
    case(64, 16, 0, 32, which=which, n=n)
