"""Dev harness (not a pytest): Winograd F(2x2,3x3) conv kernel vs float64 torch CPU and vs the direct MFMA kernel,
plus timing.  python tools/dev_wino.py [time]"""
import ctypes, sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, '.')
from scm_gan_amd import _lib as L

lib = L.lib()
dev = 'cuda'


def pack(w, wino):
    cout, cin = w.shape[0], w.shape[1]
    if wino:
        nf, nd = lib.scm_conv3x3_pack_wino_floats(cin, cout), lib.scm_conv3x3_pack_wino_floats(cout, cin)
        fp = torch.empty(nf, device=dev); dp = torch.empty(nd, device=dev)
        L.check(lib.scm_pack_conv3x3_weights_wino(L.ptr(w), cin, cout, 0, 0, 0, L.ptr(fp), L.ptr(dp), L.stream_ptr()), "pack")
    else:
        nf, nd = lib.scm_conv3x3_pack_floats(cin, cout), lib.scm_conv3x3_pack_floats(cout, cin)
        fp = torch.empty(nf, device=dev); dp = torch.empty(nd, device=dev)
        L.check(lib.scm_pack_conv3x3_weights(L.ptr(w), cin, cout, 0, 0, 0, L.ptr(fp), L.ptr(dp), L.stream_ptr()), "pack")
    return fp, dp


def conv(wino, x1, x2, wp, cout, pad, act, bias=None, sbias=None, scale=None, gate1=None, gate2=None, gate_act=0,
         co1=None, mask_out=None, mask_in=None, y1_acc=None):
    B, c1, H, W = x1.shape
    c2 = 0 if x2 is None else x2.shape[1]
    co1 = cout if co1 is None else co1
    y1 = torch.empty(B, co1, H, W, device=dev) if y1_acc is None else y1_acc
    y2 = torch.empty(B, cout - co1, H, W, device=dev) if co1 < cout else None
    d = L.ConvDesc(L.ptr(x1), L.ptr(x2), c1, c2, L.ptr(gate1), L.ptr(gate2), gate_act, L.ptr(wp), cout,
                   L.ptr(scale), L.ptr(bias), L.ptr(sbias), L.ptr(y1), L.ptr(y2), co1, B, H, W, pad, act,
                   None if mask_out is None else ctypes.c_void_p(mask_out.data_ptr()),
                   None if mask_in is None else ctypes.c_void_p(mask_in.data_ptr()), 0 if y1_acc is None else 1)
    fn = lib.scm_conv3x3_wino if wino else lib.scm_conv3x3
    L.check(fn(ctypes.byref(d), L.stream_ptr()), 'conv')
    return y1, y2


def ref_conv(x, w, b, pad, act):
    if pad == 1:
        x = F.pad(x, (1, 1, 1, 1), mode='circular'); y = F.conv2d(x, w, b)
    else:
        y = F.conv2d(x, w, b, padding=1)
    if act == 1: y = F.leaky_relu(y)
    if act == 2: y = torch.sigmoid(y)
    return y


def check():
    torch.manual_seed(0)
    ok = True
    for (B, c1, c2, cout, H, W, pad, act) in [
            (2, 32, 0, 64, 16, 16, 1, 1), (2, 128, 128, 128, 16, 16, 1, 1), (2, 9, 0, 128, 16, 16, 0, 1),
            (1, 16, 0, 64, 20, 12, 0, 1), (3, 64, 0, 48, 20, 12, 0, 0), (1, 128, 0, 128, 64, 64, 1, 1),
            (1, 21, 0, 128, 84, 84, 1, 1), (1, 8, 0, 256, 16, 16, 1, 0), (1, 128, 0, 128, 128, 128, 0, 1),
            (2, 40, 0, 70, 9, 8, 1, 1), (2, 40, 0, 70, 9, 8, 0, 2), (1, 24, 0, 64, 33, 22, 1, 0), (2, 16, 8, 100, 6, 6, 0, 1),
            (1, 12, 0, 64, 2, 2, 1, 0), (1, 12, 0, 64, 3, 6, 0, 0), (2, 128, 0, 128, 32, 32, 1, 1), (1, 64, 0, 64, 10, 100, 0, 1)]:
        x1 = torch.randn(B, c1, H, W); x2 = torch.randn(B, c2, H, W) if c2 else None
        w = torch.randn(cout, c1 + c2, 3, 3) * 0.05; b = torch.randn(cout); sb = torch.randn(B, cout) * 0.3
        sc = torch.tensor([0.7])
        xin = x1 if x2 is None else torch.cat([x1, x2], 1)
        yr = ref_conv(xin.double(), w.double() * 0.7, b.double(), pad, 0) + sb.double()[:, :, None, None]
        if act == 1: yr = F.leaky_relu(yr)
        if act == 2: yr = torch.sigmoid(yr)
        out = {}
        for wino in (1, 0):
            fp, dp = pack(w.to(dev), wino)
            co1 = 128 if cout == 256 else None
            mask = torch.zeros(B, (cout + 31) // 32, H, W, device=dev, dtype=torch.int32) if (act == 1 and cout > 32) else None
            y1, y2 = conv(wino, x1.to(dev), None if x2 is None else x2.to(dev), fp, cout, pad, act, bias=b.to(dev),
                          sbias=sb.to(dev), scale=sc.to(dev), co1=co1, mask_out=mask)
            y = y1 if y2 is None else torch.cat([y1, y2], 1)
            out[wino] = ((y.cpu().double() - yr).abs().max().item(), mask, y)
        msg = ''
        if out[1][1] is not None:
            # the 1-bit gate must equal sign(y) of the SAME kernel's output
            y = out[1][2]
            bits = (y > 0).to(torch.int32)
            nw = (cout + 31) // 32
            padc = nw * 32 - cout
            if padc: bits = torch.cat([bits, torch.zeros(B, padc, H, W, device=dev, dtype=torch.int32)], 1)
            words = (bits.view(B, nw, 32, H, W).to(torch.int64) << torch.arange(32, device=dev).view(1, 1, 32, 1, 1)).sum(2)
            got = out[1][1].to(torch.int64) & 0xffffffff
            if padc:
                keep = (1 << (32 - padc)) - 1
                got[:, -1] &= keep
            mok = bool((got == words).all())
            msg = ' mask %s' % ('ok' if mok else 'BAD')
            ok &= mok
        print('fwd', (B, c1, c2, cout, H, W, pad, act), 'err wino %.3e direct %.3e%s' % (out[1][0], out[0][0], msg))
        ok &= out[1][0] < 2e-4
        # dgrad: dx = conv(dy * act'(gate), dgrad pack), split into dx1/dx2, float gate and mask gate, accumulate
        xin_r = xin.double().requires_grad_(True)
        yr2 = ref_conv(xin_r, w.double(), None, pad, 1)
        dy = torch.randn_like(yr2)
        yr2.backward(dy)
        yf = yr2.detach().float().to(dev)
        ctot = c1 + c2
        for mode in ('float', 'mask'):
            if ctot <= 16: continue      # (a <= 16-channel gradient is the 16-row form's: ungated only; see check_w16)
            errs = []
            for wino in (1, 0):
                fp, dp = pack(w.to(dev), wino)
                mask_in = None
                if mode == 'mask':
                    if cout <= 32: continue
                    nw = (cout + 31) // 32
                    bits = (yf > 0).to(torch.int64)
                    padc = nw * 32 - cout
                    if padc: bits = torch.cat([bits, torch.zeros(B, padc, H, W, device=dev, dtype=torch.int64)], 1)
                    words = (bits.view(B, nw, 32, H, W) << torch.arange(32, device=dev).view(1, 1, 32, 1, 1)).sum(2)
                    words = torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32).contiguous()
                    mask_in = words
                base = torch.randn(B, c1, H, W, device=dev) if c2 else None
                acc = None if base is None else base.clone()
                dx1, dx2 = conv(wino, dy.float().to(dev), None, dp, ctot, pad, 0, gate1=yf, gate_act=1,
                                co1=c1 if c2 else None, mask_in=mask_in, y1_acc=acc)
                if base is not None: dx1 = dx1 - base
                dx = dx1 if dx2 is None else torch.cat([dx1, dx2], 1)
                errs.append((dx.cpu().double() - xin_r.grad).abs().max().item())
            if errs:
                print('  dgrad[%s] err wino %.3e direct %.3e' % (mode, errs[0], errs[1]))
                ok &= errs[0] < 5e-4
    print('ALL OK' if ok else 'FAILED')
    return ok


def timing():
    for (B, c1, c2, cout, H, W) in [(64, 128, 0, 128, 64, 64), (64, 128, 128, 128, 64, 64), (64, 128, 0, 256, 64, 64),
                                    (64, 16, 0, 128, 64, 64), (64, 32, 0, 128, 64, 64), (8, 128, 0, 128, 64, 64),
                                    (16, 128, 0, 128, 84, 84), (16, 128, 0, 128, 128, 128)]:
        x1 = torch.randn(B, c1, H, W, device=dev); x2 = torch.randn(B, c2, H, W, device=dev) if c2 else None
        w = torch.randn(cout, c1 + c2, 3, 3, device=dev) * 0.05; b = torch.randn(cout, device=dev)
        res = []
        for wino in (1, 0):
            fp, dp = pack(w, wino)
            co1 = 128 if cout == 256 else None
            for _ in range(3): conv(wino, x1, x2, fp, cout, 1, 1, bias=b, co1=co1)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 10
            s.record()
            for _ in range(n): conv(wino, x1, x2, fp, cout, 1, 1, bias=b, co1=co1)
            e.record(); torch.cuda.synchronize()
            res.append(s.elapsed_time(e) / n * 1e3)
        fl = 2.0 * B * H * W * cout * (c1 + c2) * 9
        print('time', (B, c1, c2, cout, H, W), 'wino %.1f us (%.1f TF algorithmic)  direct %.1f us (%.1f TF)  x%.2f' % (
            res[0], fl / res[0] / 1e6, res[1], fl / res[1] / 1e6, res[1] / res[0]))


if __name__ == '__main__' and 'wgrad' not in sys.argv and 'w16' not in sys.argv and 'wg16' not in sys.argv:
    ok = check()
    if ok or 'time' in sys.argv:
        timing()


# ---------------------------------------------------------------- weight gradient
def wgrad(wino, x1, x2, dy, gate, gate_act, pad, mask=None, want_csum=True):
    B, c1, H, W = x1.shape; c2 = 0 if x2 is None else x2.shape[1]; cout = dy.shape[1]; cin = c1 + c2
    pf = ctypes.c_size_t(); cf = ctypes.c_size_t()
    L.check(lib.scm_conv3x3_wgrad_workspace(B, H, W, cin, cout, ctypes.byref(pf), ctypes.byref(cf)), 'ws')
    part = torch.empty(pf.value, device=dev); cs = torch.empty(cf.value, device=dev)
    dw = torch.empty((cout, cin, 3, 3), device=dev)
    db = torch.empty(cout, device=dev); csum = torch.empty(B, cout, device=dev)
    d = L.WgradDesc(L.ptr(x1), L.ptr(x2), c1, c2, L.ptr(dy), L.ptr(gate), gate_act, cout, L.ptr(dw), L.ptr(db),
                    L.ptr(csum) if want_csum else None, L.ptr(part), L.ptr(cs), B, H, W, pad, 0, 0, 0, 0, None, None,
                    None if mask is None else ctypes.c_void_p(mask.data_ptr()))
    fn = lib.scm_conv3x3_wgrad_wino if wino else lib.scm_conv3x3_wgrad
    L.check(fn(ctypes.byref(d), L.stream_ptr()), 'wgrad')
    return dw, db, csum


def mask_of(y):
    B, cout, H, W = y.shape
    nw = (cout + 31) // 32
    bits = (y > 0).to(torch.int64)
    padc = nw * 32 - cout
    if padc: bits = torch.cat([bits, torch.zeros(B, padc, H, W, device=y.device, dtype=torch.int64)], 1)
    words = (bits.view(B, nw, 32, H, W) << torch.arange(32, device=y.device).view(1, 1, 32, 1, 1)).sum(2)
    return torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32).contiguous()


def check_wgrad():
    torch.manual_seed(1)
    ok = True
    for (B, c1, c2, cout, H, W, pad, act) in [
            (2, 64, 0, 64, 16, 16, 1, 1), (2, 128, 128, 128, 16, 16, 1, 1), (2, 128, 0, 128, 16, 16, 0, 1),
            (1, 64, 0, 64, 20, 12, 0, 1), (3, 64, 0, 48, 20, 12, 0, 0), (1, 128, 0, 128, 64, 64, 1, 1),
            (1, 100, 0, 128, 84, 84, 1, 1), (1, 128, 0, 128, 128, 128, 0, 1), (2, 70, 0, 70, 9, 8, 1, 1),
            (2, 70, 0, 70, 9, 8, 0, 2), (1, 64, 0, 64, 33, 24, 1, 0), (2, 64, 64, 100, 6, 8, 0, 1),
            (1, 64, 0, 64, 2, 4, 1, 0), (4, 128, 0, 128, 32, 32, 1, 1), (1, 64, 0, 64, 10, 100, 0, 1)]:
        x1 = torch.randn(B, c1, H, W); x2 = torch.randn(B, c2, H, W) if c2 else None
        w = (torch.randn(cout, c1 + c2, 3, 3) * 0.05).double().requires_grad_(True)
        bb = torch.randn(cout).double().requires_grad_(True)
        xin = (x1 if x2 is None else torch.cat([x1, x2], 1)).double()
        y = ref_conv(xin, w, bb, pad, act)
        dy = torch.randn_like(y)
        y.backward(dy)
        yf = y.detach().float().to(dev)
        res = {}
        for mode in (['float', 'mask'] if (act == 1 and cout > 32) else ['float']):
            for wino in (1, 0):
                dw, db, csum = wgrad(wino, x1.to(dev), None if x2 is None else x2.to(dev), dy.float().to(dev), yf, act, pad,
                                     mask=mask_of(yf) if mode == 'mask' else None)
                sc = max(1.0, w.grad.abs().max().item())
                e1 = (dw.cpu().double() - w.grad).abs().max().item() / sc
                e2 = (db.cpu().double() - bb.grad).abs().max().item() / max(1.0, bb.grad.abs().max().item())
                e3 = (csum.sum(0).cpu().double() - bb.grad).abs().max().item() / max(1.0, bb.grad.abs().max().item())
                res[(mode, wino)] = (e1, e2, e3)
        line = ' '.join('%s/%s dw %.1e db %.1e cs %.1e' % (m, 'wino' if wn else 'direct', *res[(m, wn)]) for (m, wn) in res)
        print('wgrad', (B, c1, c2, cout, H, W, pad, act), line)
        for (m, wn), (e1, e2, e3) in res.items():
            if wn: ok &= e1 < 1e-5 and e2 < 1e-5 and e3 < 1e-5
    print('WGRAD ALL OK' if ok else 'WGRAD FAILED')
    return ok


def timing_wgrad():
    for (B, c1, c2, cout, H, W) in [(64, 128, 0, 128, 64, 64), (64, 128, 128, 128, 64, 64), (8, 128, 0, 128, 64, 64),
                                    (16, 128, 0, 128, 84, 84), (16, 128, 0, 128, 128, 128)]:
        x1 = torch.randn(B, c1, H, W, device=dev); x2 = torch.randn(B, c2, H, W, device=dev) if c2 else None
        dy = torch.randn(B, cout, H, W, device=dev); g = torch.randn(B, cout, H, W, device=dev)
        m = mask_of(g)
        res = []
        for wino in (1, 0):
            for _ in range(3): wgrad(wino, x1, x2, dy, g, 1, 1, mask=m)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 10
            s.record()
            for _ in range(n): wgrad(wino, x1, x2, dy, g, 1, 1, mask=m)
            e.record(); torch.cuda.synchronize()
            res.append(s.elapsed_time(e) / n * 1e3)
        fl = 2.0 * B * H * W * cout * (c1 + c2) * 9
        print('wgrad time', (B, c1, c2, cout, H, W), 'wino %.1f us (%.1f TF algorithmic)  direct %.1f us (%.1f TF)  x%.2f' % (
            res[0], fl / res[0] / 1e6, res[1], fl / res[1] / 1e6, res[1] / res[0]))


if __name__ == '__main__' and 'wgrad' in sys.argv:
    ok = check_wgrad()
    if ok or 'time' in sys.argv:
        timing_wgrad()


# ---------------------------------------------------------------- 16-output-channel form
def check_w16():
    torch.manual_seed(2)
    ok = True
    for (B, c1, c2, cout, H, W, pad, act) in [
            (2, 128, 128, 16, 16, 16, 1, 2), (2, 128, 0, 16, 16, 16, 0, 2), (1, 40, 0, 16, 20, 12, 1, 0), (2, 64, 0, 3, 9, 8, 0, 1),
            (1, 128, 128, 16, 64, 64, 1, 2), (1, 32, 0, 16, 128, 128, 0, 0), (1, 32, 8, 12, 33, 22, 1, 2), (1, 48, 0, 16, 84, 84, 1, 0),
            (3, 128, 0, 16, 32, 32, 1, 0)]:
        if not lib.scm_conv3x3_wino16_serves(H, W):
            print('w16', (H, W), 'not served'); continue
        x1 = torch.randn(B, c1, H, W); x2 = torch.randn(B, c2, H, W) if c2 else None
        w = torch.randn(cout, c1 + c2, 3, 3) * 0.05; b = torch.randn(cout); sc = torch.tensor([0.7])
        xin = x1 if x2 is None else torch.cat([x1, x2], 1)
        yr = ref_conv(xin.double(), w.double() * 0.7, b.double(), pad, act)
        errs = []
        for wino in (1, 0):
            fp, dp = pack(w.to(dev), wino)
            y, _ = conv(wino, x1.to(dev), None if x2 is None else x2.to(dev), fp, cout, pad, act, bias=b.to(dev), scale=sc.to(dev))
            errs.append((y.cpu().double() - yr).abs().max().item())
        # input gradient of a (c1+c2 <- cout) layer seen from the other side: dy has c1 channels, dx has cout <= 16
        w2 = torch.randn(c1, cout, 3, 3) * 0.05           # layer cout -> c1
        xr = torch.randn(B, cout, H, W).double().requires_grad_(True)
        y2 = ref_conv(xr, w2.double(), None, pad, 0)
        dy = torch.randn_like(y2)
        y2.backward(dy)
        e2 = []
        for wino in (1, 0):
            fp, dp = pack(w2.to(dev), wino)
            dx, _ = conv(wino, dy.float().to(dev), None, dp, cout, pad, 0)
            e2.append((dx.cpu().double() - xr.grad).abs().max().item())
        print('w16', (B, c1, c2, cout, H, W, pad, act), 'fwd err wino %.3e direct %.3e | dgrad err wino %.3e direct %.3e' % (errs[0], errs[1], e2[0], e2[1]))
        ok &= errs[0] < 2e-5 and e2[0] < 2e-5
    print('W16 ALL OK' if ok else 'W16 FAILED')
    return ok


def timing_w16():
    for (B, c1, c2, cout, H, W, act) in [(64, 128, 128, 16, 64, 64, 2), (64, 128, 0, 16, 64, 64, 0), (8, 128, 128, 16, 64, 64, 2),
                                         (16, 128, 128, 16, 128, 128, 2)]:
        x1 = torch.randn(B, c1, H, W, device=dev); x2 = torch.randn(B, c2, H, W, device=dev) if c2 else None
        w = torch.randn(cout, c1 + c2, 3, 3, device=dev) * 0.05; b = torch.randn(cout, device=dev)
        res = []
        for wino in (1, 0):
            fp, dp = pack(w, wino)
            for _ in range(3): conv(wino, x1, x2, fp, cout, 1, act, bias=b)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 10
            s.record()
            for _ in range(n): conv(wino, x1, x2, fp, cout, 1, act, bias=b)
            e.record(); torch.cuda.synchronize()
            res.append(s.elapsed_time(e) / n * 1e3)
        fl = 2.0 * B * H * W * cout * (c1 + c2) * 9
        print('w16 time', (B, c1, c2, cout, H, W), 'wino %.1f us (%.1f TF algorithmic)  direct %.1f us (%.1f TF)  x%.2f' % (
            res[0], fl / res[0] / 1e6, res[1], fl / res[1] / 1e6, res[1] / res[0]))


if __name__ == '__main__' and 'w16' in sys.argv:
    ok = check_w16()
    if ok or 'time' in sys.argv:
        timing_w16()


# ---------------------------------------------------------------- weight gradient, 16-channel-side layers
def check_wg16():
    torch.manual_seed(3)
    ok = True
    for (B, c1, c2, cout, H, W, pad, act) in [
            (2, 128, 128, 16, 64, 64, 1, 2), (2, 16, 0, 128, 64, 64, 1, 0), (2, 128, 0, 16, 32, 32, 0, 2),
            (1, 16, 0, 128, 32, 64, 0, 1), (3, 40, 0, 9, 8, 32, 1, 0), (2, 9, 0, 70, 6, 32, 1, 1),
            (1, 64, 32, 12, 16, 128, 1, 1), (2, 16, 0, 16, 34, 32, 0, 0), (1, 128, 0, 16, 128, 128, 1, 2),
            (5, 16, 0, 128, 64, 64, 1, 0), (2, 128, 0, 16, 20, 12, 1, 0)]:
        x1 = torch.randn(B, c1, H, W); x2 = torch.randn(B, c2, H, W) if c2 else None
        w = (torch.randn(cout, c1 + c2, 3, 3) * 0.05).double().requires_grad_(True)
        bb = torch.randn(cout).double().requires_grad_(True)
        xin = (x1 if x2 is None else torch.cat([x1, x2], 1)).double()
        y = ref_conv(xin, w, bb, pad, act)
        dy = torch.randn_like(y)
        y.backward(dy)
        yf = y.detach().float().to(dev)
        res = {}
        for wino in (1, 0):
            dw, db, csum = wgrad(wino, x1.to(dev), None if x2 is None else x2.to(dev), dy.float().to(dev), yf if act else None,
                                 act, pad)
            sc = max(1.0, w.grad.abs().max().item())
            e1 = (dw.cpu().double() - w.grad).abs().max().item() / sc
            e2 = (db.cpu().double() - bb.grad).abs().max().item() / max(1.0, bb.grad.abs().max().item())
            e3 = (csum.sum(0).cpu().double() - bb.grad).abs().max().item() / max(1.0, bb.grad.abs().max().item())
            res[wino] = (e1, e2, e3)
        print('wg16', (B, c1, c2, cout, H, W, pad, act), ' '.join('%s dw %.1e db %.1e cs %.1e' % ('wino' if wn else 'direct', *res[wn]) for wn in res))
        ok &= all(e < 1e-5 for e in res[1])
    print('WG16 ALL OK' if ok else 'WG16 FAILED')
    return ok


def timing_wg16():
    for (B, c1, c2, cout, H, W, act) in [(64, 128, 128, 16, 64, 64, 2), (64, 16, 0, 128, 64, 64, 0), (64, 128, 0, 16, 64, 64, 2),
                                         (16, 128, 128, 16, 128, 128, 2), (8, 128, 128, 16, 64, 64, 2)]:
        x1 = torch.randn(B, c1, H, W, device=dev); x2 = torch.randn(B, c2, H, W, device=dev) if c2 else None
        dy = torch.randn(B, cout, H, W, device=dev); g = torch.rand(B, cout, H, W, device=dev) if act else None
        res = []
        for wino in (1, 0):
            for _ in range(3): wgrad(wino, x1, x2, dy, g, act, 1)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 10
            s.record()
            for _ in range(n): wgrad(wino, x1, x2, dy, g, act, 1)
            e.record(); torch.cuda.synchronize()
            res.append(s.elapsed_time(e) / n * 1e3)
        fl = 2.0 * B * H * W * cout * (c1 + c2) * 9
        print('wg16 time', (B, c1, c2, cout, H, W), 'wino %.1f us (%.1f TF algorithmic)  direct %.1f us (%.1f TF)  x%.2f (incl. reduce)' % (
            res[0], fl / res[0] / 1e6, res[1], fl / res[1] / 1e6, res[1] / res[0]))


if __name__ == '__main__' and 'wg16' in sys.argv:
    ok = check_wg16()
    if ok or 'time' in sys.argv:
        timing_wg16()
