"""Would running the weight-gradient kernel of a layer on a second stream, next to the input-gradient conv of
the same layer, hide the launch ramps?  Chain of 16 (dgrad, wgrad) pairs at 128 -> 128, B=64, 64x64 (the default Winograd kernels)."""
import ctypes, sys, torch
sys.path.insert(0, '.')
from scm_gan_amd import _lib as L, ops
lib = L.lib()
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
B, C, H, W = 64, 128, 64, 64
x = torch.randn(B, C, H, W, device='cuda'); w = torch.randn(C, C, 3, 3, device='cuda') * 0.05
gy = [torch.randn(B, C, H, W, device='cuda') * 1e-4 for _ in range(2)]
y = torch.randn(B, C, H, W, device='cuda')
mask = torch.zeros(B, C // 32, H, W, device='cuda', dtype=torch.int32)
fp, dp = ops.pack_conv3x3(w, C, C)[:2]
pf, cf = ctypes.c_size_t(), ctypes.c_size_t()
lib.scm_conv3x3_wgrad_workspace(B, H, W, C, C, ctypes.byref(pf), ctypes.byref(cf))
part = [torch.empty(pf.value, device='cuda') for _ in range(2)]; cws = [torch.empty(cf.value, device='cuda') for _ in range(2)]
dw = torch.empty_like(w); db = torch.empty(C, device='cuda')
side = torch.cuda.Stream()
def dgrad(src, dst):
    # (as the training step launches it: ungated input, the destination's gate words applied to what it writes)
    d = L.ConvDesc(P(src), None, C, 0, None, None, 0, P(dp), C, None, None, None, P(dst), None, C, B, H, W, 1, 0, None, None, 0, P(mask), None)
    L.check(lib.scm_conv3x3_wino(ctypes.byref(d), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 'dgrad')
def wgrad(g, k):
    d = L.WgradDesc(P(x), None, C, 0, P(g), None, 0, C, P(dw), P(db), None, P(part[k]), P(cws[k]), B, H, W, 1, 0, 0, 0, 0, None, None, None)
    L.check(lib.scm_conv3x3_wgrad_wino(ctypes.byref(d), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 'wgrad')
def chain(two):
    main = torch.cuda.current_stream()
    for i in range(16):
        g = gy[i & 1]; nxt = gy[(i + 1) & 1]
        if two:
            side.wait_stream(main)                  # g is ready
            with torch.cuda.stream(side):
                wgrad(g, i & 1)
            dgrad(g, nxt)
            # the next dgrad overwrites g two layers later: make main wait for this wgrad before that
            if i >= 1: pass
            main.wait_stream(side) if (i & 1) else None
        else:
            wgrad(g, i & 1)
            dgrad(g, nxt)
    main.wait_stream(side)
for two in (False, True, False, True):
    for _ in range(2): chain(two)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): chain(two)
    e.record(); torch.cuda.synchronize()
    print("two streams" if two else "one stream ", "%.1f us per (dgrad, wgrad) pair" % (s.elapsed_time(e) / 5 / 16 * 1e3))
