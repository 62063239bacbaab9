"""128->128, B=64, 64x64 launches of the bf16x3-split kernels as the training step issues them (for rocprofv3 passes)."""
import ctypes, sys, torch
sys.path.insert(0, '.')
from scm_gan_amd import _lib as L, ops
lib = L.lib()
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B, C, H, W = 64, 128, 64, 64
torch.manual_seed(0)
ops.set_compute_precision('fp32x3')
x = torch.randn(B, C, H, W, device='cuda'); w = torch.randn(C, C, 3, 3, device='cuda') * 0.05
bias = torch.zeros(C, device='cuda'); gy = torch.randn(B, C, H, W, device='cuda') * 1e-4
pf, cf = ctypes.c_size_t(), ctypes.c_size_t()
lib.scm_conv3x3_wgrad_workspace(B, H, W, C, C, ctypes.byref(pf), ctypes.byref(cf))
part = torch.empty(pf.value, device='cuda'); cws = torch.empty(cf.value, device='cuda')
dw = torch.empty_like(w); db = torch.empty(C, device='cuda')
packs = ops.pack_conv3x3(w, C, C)
mask = torch.zeros(B, C // 32, H, W, device='cuda', dtype=torch.int32)
for _ in range(16):
    y, _ = ops._conv3x3_raw(x, None, None, None, 0, packs[0], C, None, bias, None, C, 1, ops.ACT_LRELU, packs[2], mask_out=mask)
    dx, _ = ops._conv3x3_raw(gy, None, y, None, ops.ACT_LRELU, packs[1], C, None, None, None, C, 1, ops.ACT_NONE, packs[3], mask_in=mask)
    d = L.WgradDesc(P(x), None, C, 0, P(gy), P(y), ops.ACT_LRELU, C, P(dw), P(db), None, P(part), P(cws), B, H, W, 1, 0, 0, 0, 0, None, None, P(mask))
    L.check(lib.scm_conv3x3_wgrad_h16(ctypes.byref(d), ops.PREC_BF16X3, S()), 'wgrad_x3')
torch.cuda.synchronize()
