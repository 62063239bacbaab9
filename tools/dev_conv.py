"""Dev harness (not a pytest): conv3x3 MFMA kernel vs torch CPU, plus timing."""
import ctypes, sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, '.')
from scm_gan_amd import _lib as L

lib = L.lib()
dev = 'cuda'

def pack(w, transposed=False):
    if transposed:
        cin, cout = w.shape[0], w.shape[1]
    else:
        cout, cin = w.shape[0], w.shape[1]
    nf = lib.scm_conv3x3_pack_floats(cin, cout)
    nd = lib.scm_conv3x3_pack_floats(cout, cin)
    fp = torch.empty(nf, device=dev); dp = torch.empty(nd, device=dev)
    L.check(lib.scm_pack_conv3x3_weights(L.ptr(w), cin, cout, int(transposed), 0, 0, L.ptr(fp), L.ptr(dp), L.stream_ptr()), "pack")
    return fp, dp

def conv(x1, x2, wp, cout, pad, act, bias=None, sbias=None, scale=None, gate1=None, gate2=None, gate_act=0, co1=None):
    B, c1, H, W = x1.shape
    c2 = 0 if x2 is None else x2.shape[1]
    co1 = cout if co1 is None else co1
    y1 = torch.empty(B, co1, H, W, device=dev)
    y2 = torch.empty(B, cout - co1, H, W, device=dev) if co1 < cout else None
    d = L.ConvDesc(L.ptr(x1), L.ptr(x2), c1, c2, L.ptr(gate1), L.ptr(gate2), gate_act, L.ptr(wp), cout,
                   L.ptr(scale), L.ptr(bias), L.ptr(sbias), L.ptr(y1), L.ptr(y2), co1, B, H, W, pad, act)
    L.check(lib.scm_conv3x3(ctypes.byref(d), L.stream_ptr()), 'conv')
    return y1, y2

def ref_conv(x, w, b, pad, act):
    if pad == 1:
        x = F.pad(x, (1, 1, 1, 1), mode='circular'); y = F.conv2d(x, w, b)
    else:
        y = F.conv2d(x, w, b, padding=1)
    if act == 1: y = F.leaky_relu(y)
    if act == 2: y = torch.sigmoid(y)
    return y

torch.manual_seed(0)
ok = True
for (B, c1, c2, cout, H, W, pad, act) in [
    (2, 16, 0, 128, 16, 16, 1, 1), (2, 128, 128, 128, 16, 16, 1, 1), (2, 9, 0, 128, 16, 16, 0, 1),
    (2, 128, 0, 16, 16, 16, 0, 2), (2, 128, 128, 16, 16, 16, 1, 2), (1, 16, 0, 64, 20, 12, 0, 1),
    (3, 64, 0, 48, 20, 12, 0, 0), (1, 128, 0, 128, 64, 64, 1, 1), (1, 21, 0, 128, 84, 84, 1, 1),
    (1, 8, 0, 256, 16, 16, 1, 0), (1, 128, 0, 128, 128, 128, 0, 1)]:
    x1 = torch.randn(B, c1, H, W); x2 = torch.randn(B, c2, H, W) if c2 else None
    w = torch.randn(cout, c1 + c2, 3, 3) * 0.05; b = torch.randn(cout)
    xin = x1 if x2 is None else torch.cat([x1, x2], 1)
    yr = ref_conv(xin.double(), w.double(), b.double(), pad, act).float()
    fp, dp = pack(w.to(dev))
    co1 = 128 if cout == 256 else None
    y1, y2 = conv(x1.to(dev), None if x2 is None else x2.to(dev), fp, cout, pad, act, bias=b.to(dev), co1=co1)
    y = y1 if y2 is None else torch.cat([y1, y2], 1)
    err = (y.cpu() - yr).abs().max().item()
    print('fwd', (B, c1, c2, cout, H, W, pad, act), 'maxerr %.3e' % err)
    ok &= err < 2e-4
    # dgrad check: dx = conv(dy, dgrad_pack) against autograd (no act)
    xin_r = xin.double().requires_grad_(True)
    yr2 = ref_conv(xin_r, w.double(), None, pad, 0)
    dy = torch.randn_like(yr2)
    yr2.backward(dy)
    dx1, dx2 = conv(dy.float().to(dev), None, dp, c1 + c2, pad, 0, co1=c1 if c2 else None)
    dx = dx1 if dx2 is None else torch.cat([dx1, dx2], 1)
    err = (dx.cpu() - xin_r.grad.float()).abs().max().item()
    print('  dgrad maxerr %.3e' % err)
    ok &= err < 5e-4
print('ALL OK' if ok else 'FAILED')

# timing: Transition conv2 at cfg2 (B=64, 128->128, 64x64, circular, lrelu)
for (B, c1, c2, cout, H, W) in [(64, 128, 0, 128, 64, 64), (64, 128, 128, 128, 64, 64), (64, 128, 128, 16, 64, 64), (64, 16, 0, 128, 64, 64)]:
    x1 = torch.randn(B, c1, H, W, device=dev); x2 = torch.randn(B, c2, H, W, device=dev) if c2 else None
    w = torch.randn(cout, c1 + c2, 3, 3, device=dev) * 0.05; b = torch.randn(cout, device=dev)
    fp, dp = pack(w)
    for _ in range(3): conv(x1, x2, fp, cout, 1, 1, bias=b)
    torch.cuda.synchronize(); t0 = time.time(); n = 10
    for _ in range(n): conv(x1, x2, fp, cout, 1, 1, bias=b)
    torch.cuda.synchronize(); dt = (time.time() - t0) / n
    fl = 2.0 * B * H * W * cout * (c1 + c2) * 9
    print('time', (B, c1, c2, cout, H, W), '%.3f ms  %.1f TFLOP/s' % (dt * 1e3, fl / dt / 1e12))

# ---------------- wgrad
def wgrad(x1, x2, dy, gate, gate_act, pad, transposed=False):
    B, c1, H, W = x1.shape; c2 = 0 if x2 is None else x2.shape[1]; cout = dy.shape[1]; cin = c1 + c2
    pf = ctypes.c_size_t(); cf = ctypes.c_size_t()
    L.check(lib.scm_conv3x3_wgrad_workspace(B, H, W, cin, cout, ctypes.byref(pf), ctypes.byref(cf)), 'ws')
    part = torch.empty(pf.value, device=dev); cs = torch.empty(cf.value, device=dev)
    dw = torch.empty((cin, cout, 3, 3) if transposed else (cout, cin, 3, 3), device=dev)
    db = torch.empty(cout, device=dev); csum = torch.empty(B, cout, device=dev)
    d = L.WgradDesc(L.ptr(x1), L.ptr(x2), c1, c2, L.ptr(dy), L.ptr(gate), gate_act, cout, L.ptr(dw), L.ptr(db), L.ptr(csum),
                    L.ptr(part), L.ptr(cs), B, H, W, pad, int(transposed), 0, 0, 0, None, None)
    L.check(lib.scm_conv3x3_wgrad(ctypes.byref(d), L.stream_ptr()), 'wgrad')
    return dw, db, csum

ok = True
for (B, c1, c2, cout, H, W, pad, act) in [
    (2, 16, 0, 128, 16, 16, 1, 1), (2, 128, 128, 128, 16, 16, 1, 1), (2, 9, 0, 128, 16, 16, 0, 1),
    (2, 128, 0, 16, 16, 16, 0, 2), (2, 128, 128, 16, 16, 16, 1, 2), (1, 16, 0, 64, 20, 12, 0, 1),
    (3, 64, 0, 48, 20, 13, 0, 0), (1, 128, 0, 128, 64, 64, 1, 1), (1, 21, 0, 128, 84, 84, 1, 1),
    (2, 40, 0, 70, 9, 7, 1, 1), (1, 128, 0, 128, 128, 128, 0, 1)]:
    x1 = torch.randn(B, c1, H, W); x2 = torch.randn(B, c2, H, W) if c2 else None
    w = (torch.randn(cout, c1 + c2, 3, 3) * 0.05).double().requires_grad_(True); bb = torch.randn(cout).double().requires_grad_(True)
    xin = (x1 if x2 is None else torch.cat([x1, x2], 1)).double()
    y = ref_conv(xin, w, bb, pad, act)
    dy = torch.randn_like(y)
    y.backward(dy)
    dw, db, csum = wgrad(x1.to(dev), None if x2 is None else x2.to(dev), dy.float().to(dev), y.detach().float().to(dev), act, pad)
    e1 = (dw.cpu() - w.grad.float()).abs().max().item() / max(1.0, w.grad.abs().max().item())
    e2 = (db.cpu() - bb.grad.float()).abs().max().item() / max(1.0, bb.grad.abs().max().item())
    e3 = (csum.sum(0).cpu() - bb.grad.float()).abs().max().item() / max(1.0, bb.grad.abs().max().item())
    print('wgrad', (B, c1, c2, cout, H, W, pad, act), 'rel err dw %.2e db %.2e csum %.2e' % (e1, e2, e3))
    ok &= e1 < 1e-4 and e2 < 1e-4 and e3 < 1e-4
print('WGRAD ALL OK' if ok else 'WGRAD FAILED')
for (B, c1, c2, cout, H, W) in [(64, 128, 0, 128, 64, 64), (64, 128, 128, 128, 64, 64), (64, 128, 128, 16, 64, 64), (64, 16, 0, 128, 64, 64)]:
    x1 = torch.randn(B, c1, H, W, device=dev); x2 = torch.randn(B, c2, H, W, device=dev) if c2 else None
    dy = torch.randn(B, cout, H, W, device=dev); g = torch.randn(B, cout, H, W, device=dev)
    for _ in range(2): wgrad(x1, x2, dy, g, 1, 1)
    torch.cuda.synchronize(); t0 = time.time(); n = 5
    for _ in range(n): wgrad(x1, x2, dy, g, 1, 1)
    torch.cuda.synchronize(); dt = (time.time() - t0) / n
    fl = 2.0 * B * H * W * cout * (c1 + c2) * 9
    print('wgrad time', (B, c1, c2, cout, H, W), '%.3f ms  %.1f TFLOP/s' % (dt * 1e3, fl / dt / 1e12))
