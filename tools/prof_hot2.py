"""128->128, B=64, 64x64 launches of the hot kernels as the training step issues them (1-bit gate mask on; since round 3
also the producer-gated forms and the 16-channel-side layers of the transition):
fp32 forward / gated dgrad / wgrad in the default Winograd form and in the direct form, their fp32-on-the-bf16-pipe (fp32x3)
and 16-bit-operand forms, on preallocated buffers (for rocprofv3 --pmc passes)."""
import ctypes, sys, torch
sys.path.insert(0, '.')
from scm_gan_amd import _lib as L, ops
lib = L.lib()
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
S = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B, C, H, W = 64, 128, 64, 64
torch.manual_seed(0)
x = torch.randn(B, C, H, W, device='cuda'); w = torch.randn(C, C, 3, 3, device='cuda') * 0.05
bias = torch.zeros(C, device='cuda'); gy = torch.randn(B, C, H, W, device='cuda') * 1e-4
pf, cf = ctypes.c_size_t(), ctypes.c_size_t()
lib.scm_conv3x3_wgrad_workspace(B, H, W, C, C, ctypes.byref(pf), ctypes.byref(cf))
part = torch.empty(pf.value, device='cuda'); cws = torch.empty(cf.value, device='cuda')
dw = torch.empty_like(w); db = torch.empty(C, device='cuda')
for mode in ('fp32', 'fp32-direct', 'fp32x3', 'fp16'):
    ops.set_conv_algo('direct' if mode == 'fp32-direct' else 'winograd')
    ops.set_compute_precision('fp32' if mode == 'fp32-direct' else mode)
    fp, dp, fprec, dprec = ops.pack_conv3x3(w, C, C)[:4]
    mask = torch.zeros(B, C // 32, H, W, device='cuda', dtype=torch.int32)
    for _ in range(3):
        if mode == 'fp32':
            # the product's form since round 3 (producer-side gating): the input gradient runs ungated and gates what it
            # WRITES with the destination's mask -- the same instantiation as the forward, so it goes first and the forward's
            # dispatch is the one the parser keeps; the weight gradient then runs ungated too (below)
            ops._conv3x3_raw(gy, None, None, None, 0, dp, C, None, None, None, C, 1, ops.ACT_NONE, dprec, ymask1=mask)
        y, _ = ops._conv3x3_raw(x, None, None, None, 0, fp, C, None, bias, None, C, 1, ops.ACT_LRELU, fprec, mask_out=mask)
        dx, _ = ops._conv3x3_raw(gy, None, y, None, ops.ACT_LRELU, dp, C, None, None, None, C, 1, ops.ACT_NONE, dprec, mask_in=mask)
        d = L.WgradDesc(P(x), None, C, 0, P(gy), P(y), ops.ACT_LRELU, C, P(dw), P(db), None, P(part), P(cws), B, H, W, 1, 0, 0, 0, 0,
                        None, None, P(mask))
        if dprec == ops.ALGO_WINO:
            L.check(lib.scm_conv3x3_wgrad_wino(ctypes.byref(d), S()), 'wgrad_wino')
        elif dprec:
            L.check(lib.scm_conv3x3_wgrad_h16(ctypes.byref(d), dprec, S()), 'wgrad_h16')
        else:
            L.check(lib.scm_conv3x3_wgrad(ctypes.byref(d), S()), 'wgrad')
        if mode == 'fp32':
            d0 = L.WgradDesc(P(x), None, C, 0, P(gy), None, 0, C, P(dw), P(db), None, P(part), P(cws), B, H, W, 1, 0, 0, 0, 0,
                             None, None, None)
            L.check(lib.scm_conv3x3_wgrad_wino(ctypes.byref(d0), S()), 'wgrad_wino ungated')
    torch.cuda.synchronize()
# the 16-channel-side layers of the transition (fp32, Winograd forms): conv6's forward (256 -> 16, sigmoid) and weight
# gradient (sigmoid gate), conv1's weight gradient (16 -> 128, ungated)
ops.set_compute_precision('fp32')
ops.set_conv_algo('winograd')
xa, xb_ = torch.randn(B, C, H, W, device='cuda'), torch.randn(B, C, H, W, device='cuda')
w6 = torch.randn(16, 2 * C, 3, 3, device='cuda') * 0.05
z16 = torch.randn(B, 16, H, W, device='cuda'); g16 = torch.randn(B, 16, H, W, device='cuda') * 1e-4
fp6 = ops.pack_conv3x3(w6, 2 * C, 16)
fp1 = ops.pack_conv3x3(torch.randn(C, 16, 3, 3, device='cuda') * 0.05, 16, C)
pf6, cf6 = ctypes.c_size_t(), ctypes.c_size_t()
lib.scm_conv3x3_wgrad_workspace(B, H, W, 2 * C, 16, ctypes.byref(pf6), ctypes.byref(cf6))
part6 = torch.empty(max(pf6.value, pf.value), device='cuda'); cws6 = torch.empty(max(cf6.value, cf.value), device='cuda')
dw6 = torch.empty_like(w6); db6 = torch.empty(16, device='cuda'); dw1 = torch.empty(C, 16, 3, 3, device='cuda'); db1 = torch.empty(C, device='cuda')
for _ in range(3):
    p6, _ = ops._conv3x3_raw(xa, xb_, None, None, 0, fp6[0], 16, None, torch.zeros(16, device='cuda'), None, 16, 1, ops.ACT_SIGMOID, fp6[2])
    d6 = L.WgradDesc(P(xa), P(xb_), C, C, P(g16), P(p6), ops.ACT_SIGMOID, 16, P(dw6), P(db6), None, P(part6), P(cws6), B, H, W, 1,
                     0, 0, 0, 0, None, None, None)
    L.check(lib.scm_conv3x3_wgrad_wino(ctypes.byref(d6), S()), 'wgrad_wino conv6')
    d1 = L.WgradDesc(P(z16), None, 16, 0, P(gy), None, 0, C, P(dw1), P(db1), None, P(part6), P(cws6), B, H, W, 1, 0, 0, 0, 0,
                     None, None, None)
    L.check(lib.scm_conv3x3_wgrad_wino(ctypes.byref(d1), S()), 'wgrad_wino conv1')
    # conv6's input gradient (16 -> 256, sigmoid gate, both outputs gated by their activations' masks) and conv1's forward
    # (16 -> 128, leaky-relu + gate words): the 16-input-channel kernel
    ops._conv3x3_raw(g16, None, p6, None, ops.ACT_SIGMOID, fp6[1], 2 * C, None, None, None, C, 1, ops.ACT_NONE, fp6[3],
                     ymask1=mask, ymask2=mask)
    ops._conv3x3_raw(z16, None, None, None, 0, fp1[0], C, None, bias, None, C, 1, ops.ACT_LRELU, fp1[2], mask_out=mask)
torch.cuda.synchronize()
# the fused decoder + BCE launches of one training iteration (fp32): forward (keeping the hidden map) and backward over
# the 8 x 64 samples of all unrolled steps at once (time-batched heads, Trainer.step); then the forward as inference /
# bench.py's decoder leg launches it (B = 64, no hidden map kept) -- the LAST dispatch of the forward kernel, the one
# tools/pmc_parse.py records
ops.set_compute_precision('fp32')
ops.set_conv_algo('winograd')
from scm_gan_amd import models as M
NT = 8
dec = M.Decoder(16, 3)
z = (torch.rand(NT * B, 16, H, W, device='cuda') < 0.5).float().requires_grad_(True)
tgt = (torch.rand(NT * B, 3, H, W, device='cuda') < 0.1).float()
for _ in range(3):
    dec.forward_loss(z, tgt).mean().backward()
torch.cuda.synchronize()
with torch.no_grad():
    for _ in range(3):
        dec.forward_loss(z[:B], tgt[:B])
torch.cuda.synchronize()
# the encoder's forward as the training step launches it (zero padding: conv1 9 -> 128 on the 16-input-channel kernel,
# conv2 / conv3 on the wide kernel's zero-padded instantiation, conv4 128 -> 16 on the 16-row kernel) -- the north star's
# "MFMA utilisation on the encoder convs" -- and its backward
enc = M.Encoder(16, 3)
xe = (torch.rand(B, 3, 3, H, W, device='cuda') < 0.1).float()
for _ in range(3):
    enc(xe).sum().backward()
torch.cuda.synchronize()
# the fused reward head, forward + backward, time-batched as the training iteration launches it (8 x 64 samples)
rp = M.RewardPredictor(16, 2)
zr = (torch.rand(NT * B, 16, H, W, device='cuda') < 0.5).float().requires_grad_(True)
for _ in range(3):
    rp(zr).sum().backward()
torch.cuda.synchronize()
# the 16-bit operand mode with the activations STORED as IEEE half (scm_conv3x3_desc.x_half / y_half; opt-in,
# ops.set_half_store): forward 128 -> 128 reading and writing half tensors, weight gradient reading a half x
ops.set_compute_precision('fp16')
fph, dph, fprech, dprech = ops.pack_conv3x3(w, C, C)[:4]
xh16 = x.half()
for _ in range(3):
    yh, _ = ops._conv3x3_raw(xh16, None, None, None, 0, fph, C, None, bias, None, C, 1, ops.ACT_LRELU, fprech, mask_out=mask, y_half=True)
    dh = L.WgradDesc(P(xh16), None, C, 0, P(gy), P(gy), ops.ACT_LRELU, C, P(dw), P(db), None, P(part), P(cws), B, H, W, 1, 0, 0, 0, 0,
                     None, None, P(mask), 1)
    L.check(lib.scm_conv3x3_wgrad_h16(ctypes.byref(dh), dprech, S()), 'wgrad_h16 half x')
torch.cuda.synchronize()
ops.set_compute_precision('fp32')
