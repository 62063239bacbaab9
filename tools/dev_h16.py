"""scm_conv3x3_h16 vs torch on the SAME rounded operands (exact up to fp32 accumulation order) + timing."""
import ctypes, sys, torch
import torch.nn.functional as F
sys.path.insert(0, '.')
from scm_gan_amd import _lib as L, ops
lib = L.lib()
dev = 'cuda'
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
RD = {1: torch.float16, 2: torch.bfloat16}


def timeit(fn, n=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def pad(x, circ):
    return F.pad(x, (1, 1, 1, 1), mode='circular') if circ else F.pad(x, (1, 1, 1, 1))


def case(B, c1, c2, cout, H=64, W=64, circ=1, check=True, n=10):
    torch.manual_seed(0)
    cin = c1 + c2
    x1 = torch.randn(B, c1, H, W, device=dev); x2 = torch.randn(B, c2, H, W, device=dev) if c2 else None
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05; b = torch.randn(cout, device=dev)
    sc = torch.tensor([0.7], device=dev)
    nb = lib.scm_conv3x3_pack_h16_bytes
    fp = torch.empty(nb(cin, cout) // 2, device=dev, dtype=torch.int16)
    dp = torch.empty(nb(cout, cin) // 2, device=dev, dtype=torch.int16)
    L.check(lib.scm_pack_conv3x3_weights_h16(P(w), cin, cout, 0, 0, 0, P(fp), 1, P(dp), 2, S()), 'pack')
    y = torch.empty(B, cout, H, W, device=dev); dy = torch.randn(B, cout, H, W, device=dev) * 1e-6
    dx1 = torch.empty(B, c1, H, W, device=dev); dx2 = torch.empty(B, c2, H, W, device=dev) if c2 else None
    d = L.ConvDesc(P(x1), P(x2), c1, c2, None, None, 0, P(fp), cout, P(sc), P(b), None, P(y), None, cout, B, H, W, circ, 1)
    L.check(lib.scm_conv3x3_h16(ctypes.byref(d), 1, S()), 'fwd')
    d2 = L.ConvDesc(P(dy), None, cout, 0, P(y), None, 1, P(dp), cin, P(sc), None, None, P(dx1), P(dx2), c1, B, H, W, circ, 0)
    L.check(lib.scm_conv3x3_h16(ctypes.byref(d2), 2, S()), 'dgrad')
    torch.cuda.synchronize()
    msg = ''
    if check:
        x = x1 if x2 is None else torch.cat([x1, x2], 1)
        xr, wr = x.half().double(), w.half().double()
        ref = F.leaky_relu(F.conv2d(pad(xr, circ), wr) * 0.7 + b.double().view(1, -1, 1, 1), 0.01)
        e1 = ((y.double() - ref).abs().max() / ref.abs().max()).item()
        full = F.leaky_relu(F.conv2d(pad(x.double(), circ), w.double()) * 0.7 + b.double().view(1, -1, 1, 1), 0.01)
        e1f = ((y.double() - full).abs().max() / full.abs().max()).item()
        g = (dy * torch.where(y > 0, torch.ones_like(y), torch.full_like(y, 0.01))).bfloat16().double()
        wb = w.bfloat16().double()
        gp = pad(g, circ)
        refd = F.conv2d(gp, wb.flip(2, 3).transpose(0, 1)) * 0.7
        got = dx1 if dx2 is None else torch.cat([dx1, dx2], 1)
        e2 = ((got.double() - refd).abs().max() / refd.abs().max()).item()
        msg = ' | err fwd %.2e (vs fp32 conv %.2e) dgrad %.2e' % (e1, e1f, e2)
    fl = 2.0 * B * H * W * cout * cin * 9
    t1 = timeit(lambda: L.check(lib.scm_conv3x3_h16(ctypes.byref(d), 1, S()), 'fwd'), n)
    t2 = timeit(lambda: L.check(lib.scm_conv3x3_h16(ctypes.byref(d2), 2, S()), 'dgrad'), n)
    byt = 4.0 * B * H * W * (cin + cout)
    print('B %3d %3d+%3d -> %3d %dx%d c%d | fwd %7.1f us %6.1f TF %5.2f TB/s | dgrad %7.1f us %6.1f TF%s'
          % (B, c1, c2, cout, H, W, circ, t1, fl / t1 / 1e6, byt / t1 / 1e6, t2, fl / t2 / 1e6, msg)); sys.stdout.flush()


if __name__ == '__main__' and 'c' in sys.argv[1:]:
    case(2, 20, 0, 40, 16, 16, 1)
    case(3, 16, 5, 128, 20, 12, 0)
    case(2, 7, 0, 3, 33, 17, 0)
    case(2, 128, 128, 16, 64, 64, 1)
    case(1, 16, 0, 64, 84, 84, 0)
    case(1, 48, 0, 128, 128, 128, 1)
    case(64, 128, 0, 128, check=False)
    case(64, 128, 128, 128, check=False)
    case(64, 128, 128, 16, check=False)
    case(64, 16, 0, 128, check=False)
    case(64, 64, 0, 3, check=False)


def wcase(B, c1, c2, cout, H=64, W=64, circ=1, check=True, n=10, gate=1):
    torch.manual_seed(1)
    cin = c1 + c2
    x1 = torch.randn(B, c1, H, W, device=dev); x2 = torch.randn(B, c2, H, W, device=dev) if c2 else None
    y = torch.randn(B, cout, H, W, device=dev); dy = torch.randn(B, cout, H, W, device=dev) * 1e-5
    pf, cf = ctypes.c_size_t(), ctypes.c_size_t()
    lib.scm_conv3x3_wgrad_workspace(B, H, W, cin, cout, ctypes.byref(pf), ctypes.byref(cf))
    part = torch.empty(pf.value, device=dev); cws = torch.empty(cf.value, device=dev)
    dw = torch.empty(cout, cin, 3, 3, device=dev); db = torch.empty(cout, device=dev); cs = torch.empty(B, cout, device=dev)
    d3 = L.WgradDesc(P(x1), P(x2), c1, c2, P(dy), P(y) if gate else None, gate, cout, P(dw), P(db), P(cs), P(part), P(cws),
                     B, H, W, circ, 0, 0, 0, 0, None, None)
    L.check(lib.scm_conv3x3_wgrad_h16(ctypes.byref(d3), 2, S()), 'wgrad16')
    torch.cuda.synchronize()
    msg = ''
    if check:
        x = x1 if x2 is None else torch.cat([x1, x2], 1)
        gfac = torch.where(y > 0, torch.ones_like(y), torch.full_like(y, 0.01)) if gate else torch.ones_like(y)
        dyp = dy * gfac
        xr, gr = x.bfloat16().double(), dyp.bfloat16().double()
        xp = pad(xr, circ)
        ref = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, device=dev)
        for ky in range(3):
            for kx in range(3):
                ref[:, :, ky, kx] = torch.einsum('bohw,bihw->oi', gr, xp[:, :, ky:ky + H, kx:kx + W])
        e = ((dw.double() - ref).abs().max() / ref.abs().max()).item()
        ecs = ((cs.double() - dyp.double().sum((2, 3))).abs().max() / dyp.double().sum((2, 3)).abs().max()).item()
        edb = ((db.double() - dyp.double().sum((0, 2, 3))).abs().max() / dyp.double().sum((0, 2, 3)).abs().max()).item()
        dw32 = torch.empty_like(dw)
        d4 = L.WgradDesc(P(x1), P(x2), c1, c2, P(dy), P(y) if gate else None, gate, cout, P(dw32), P(db), P(cs), P(part), P(cws),
                         B, H, W, circ, 0, 0, 0, 0, None, None)
        L.check(lib.scm_conv3x3_wgrad(ctypes.byref(d4), S()), 'wgrad32')
        e32 = ((dw - dw32).abs().max() / dw32.abs().max()).item()
        msg = ' | err dw %.2e (vs fp32 kernel %.2e) csum %.2e db %.2e' % (e, e32, ecs, edb)
    fl = 2.0 * B * H * W * cout * cin * 9
    t = timeit(lambda: L.check(lib.scm_conv3x3_wgrad_h16(ctypes.byref(d3), 2, S()), 'w'), n)
    print('wgrad B %3d %3d+%3d -> %3d %dx%d c%d | %7.1f us %6.1f TF%s' % (B, c1, c2, cout, H, W, circ, t, fl / t / 1e6, msg))
    sys.stdout.flush()


if __name__ == '__main__' and 'w' in sys.argv[1:]:
    wcase(2, 64, 0, 128, 16, 16, 1)
    wcase(2, 16, 0, 128, 16, 16, 1)
    wcase(3, 9, 0, 128, 20, 12, 0)
    wcase(2, 20, 5, 100, 64, 64, 1)
    wcase(64, 16, 0, 128, check=False)
    wcase(3, 40, 30, 100, 20, 12, 0)
    wcase(2, 128, 128, 128, 64, 64, 1)
    wcase(1, 64, 0, 128, 84, 84, 0, gate=0)
    wcase(1, 48, 0, 128, 128, 128, 1)
    wcase(64, 128, 0, 128, check=False)
    wcase(64, 128, 128, 128, check=False)
