"""Launch the hot conv kernels a few times (for rocprofv3 --pmc runs)."""
import ctypes, sys, torch
sys.path.insert(0, '.')
from scm_gan_amd import ops
dev = 'cuda'
B, C, H, W = 64, 128, 64, 64
torch.manual_seed(0)
x = torch.randn(B, C, H, W, device=dev); x2 = torch.randn(B, C, H, W, device=dev)
w = torch.randn(128, 128, 3, 3, device=dev) * 0.05; w2 = torch.randn(128, 256, 3, 3, device=dev) * 0.05
b = torch.zeros(128, device=dev)
fp, dp = ops.pack_conv3x3(w, 128, 128)[:2]; fp2, dp2 = ops.pack_conv3x3(w2, 256, 128)[:2]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for _ in range(n):
    y, _ = ops._conv3x3_raw(x, None, None, None, 0, fp, 128, None, b, None, 128, 1, 1)          # fwd 128->128
    y2, _ = ops._conv3x3_raw(x, x2, None, None, 0, fp2, 128, None, b, None, 128, 1, 1)          # fwd 256->128
    d1, _ = ops._conv3x3_raw(y, None, y, None, 1, dp, 128, None, None, None, 128, 1, 0)         # dgrad (gated)
cfg = {"pad": 1, "act": 1, "transposed": False, "packs": (fp, dp, 0, 0)}
xx = x.clone().requires_grad_(True); ww = w.clone().requires_grad_(True); bb = b.clone().requires_grad_(True)
for _ in range(n):
    out = ops.Conv3x3Fn.apply(xx, None, None, ww, bb, None, None, None, cfg)
    out.backward(y)
torch.cuda.synchronize()
print("done")
