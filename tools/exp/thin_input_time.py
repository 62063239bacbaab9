"""Times the 16-input-channel launches of the wide Winograd kernel (conv1's forward 16 -> 128, conv6's sigmoid-gated input
gradient 16 -> 256 with split, gated outputs) in its 128-co and 64-co workgroup forms (SCM_WINO_NO128)."""
import ctypes, os, sys, torch
sys.path.insert(0, '.')
from scm_gan_amd import _lib as L, ops
lib = L.lib()
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
B, H, W = 64, 64, 64
torch.manual_seed(0)
for cout in (128, 256):
    x = torch.randn(B, 16, H, W, device='cuda'); g = torch.rand(B, 16, H, W, device='cuda')
    w = torch.randn(cout, 16, 3, 3, device='cuda') * 0.05
    fp = ops.pack_conv3x3(w, 16, cout)
    m1 = torch.randint(-2**31, 2**31 - 1, (B, 4, H, W), device='cuda', dtype=torch.int32)
    def run():
        if cout == 128:
            return ops._conv3x3_raw(x, None, None, None, 0, fp[0], cout, None, torch.zeros(cout, device='cuda'), None, cout, 1,
                                    ops.ACT_LRELU, fp[2], mask_out=m1)
        return ops._conv3x3_raw(x, None, g, None, ops.ACT_SIGMOID, fp[0], cout, None, None, None, 128, 1, ops.ACT_NONE, fp[2],
                                ymask1=m1, ymask2=m1)
    for _ in range(3): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): run()
    e.record(); torch.cuda.synchronize()
    print('16 -> %d  SCM_WINO_NO128=%s: %.1f us' % (cout, os.environ.get('SCM_WINO_NO128', '0'), s.elapsed_time(e) / 20 * 1e3))
