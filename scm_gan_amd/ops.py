"""Operators of the hot path as torch.autograd.Functions over the C ABI
(include/scm_hip.h).  PyTorch owns memory, streams and the graph topology;
every FLOP and byte of these operators runs in libscmgan_hip.so.  There is no
fallback: tensors must be fp32 CUDA tensors and the library must be built.
"""
import ctypes
import os

import torch
from torch.autograd import Function

from . import _lib as L

PAD_ZERO, PAD_CIRCULAR = L.PAD_ZERO, L.PAD_CIRCULAR
ACT_NONE, ACT_LRELU, ACT_SIGMOID = L.ACT_NONE, L.ACT_LRELU, L.ACT_SIGMOID


# Bumped by optimizers that update parameters through raw device pointers (which
# torch's tensor version counter cannot see); part of every weight-pack cache key.
_weights_epoch = [0]


def weights_epoch():
    return _weights_epoch[0]


def bump_weights_epoch():
    _weights_epoch[0] += 1


# Device-resident per-iteration scalars of a captured training iteration (graph_step.GraphedTrainer sets them
# while it captures): 'coef' float32[1] (theta * reward_coef), 'noise_off' int64[1] (Philox stream position),
# 'adam_step' int32[1].  Empty in eager mode: the same values then travel by value.
dev_scalars = {}


def _dev(name):
    t = dev_scalars.get(name)
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _chk(t, name, half_ok=False):
    if t is None:
        return None
    if not t.is_cuda or (t.dtype != torch.float32 and not (half_ok and t.dtype == torch.float16)):
        raise RuntimeError("scm_gan_amd.%s: expected an fp32 tensor on the GPU (got %s on %s); "
                           "there is no CPU fallback" % (name, t.dtype, t.device))
    return t.contiguous()


def _new(like, *shape):
    return torch.empty(shape, device=like.device, dtype=torch.float32)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# ------------------------------------------------------------- precision
# MFMA operand precision of the 3x3 same-size convolutions.  "fp32" (default, the
# reference's arithmetic) or "fp16": forward on fp16 operands, gradient convolutions on
# bf16 operands, both with fp32 accumulation over the same fp32 tensors
# (scm_conv3x3_h16; BASELINE config 5 / SURVEY.md 8(d)).  Opt-in: parity claims and the
# headline benchmark are fp32.
#
# "fp32x3": fp32 arithmetic on the bf16 MFMA pipe (scm_conv3x3_x3): every operand split exactly into three bf16
# values, six partial products per product, fp32 accumulation.  fp32-grade results (tests/test_gpu_x3_mode.py
# measures its error against float64 next to the native kernel's) at 6/16 of the f32 MFMA cost.  Serves the forward
# and input-gradient convolutions with more than 16 output channels on frames up to 64 wide; everything else
# (weight gradients, thin layers, fused decoder / reward head) runs the native fp32 kernels in this mode too.
PREC_F32, PREC_F16, PREC_BF16, PREC_BF16X3 = L.PREC_F32, L.PREC_F16, L.PREC_BF16, L.PREC_BF16X3
# not a precision: the fp32 convolution evaluated by Winograd F(2x2, 3x3) on the f32 MFMA pipe (scm_conv3x3_wino);
# a pack tagged with it goes to that entry point
ALGO_WINO = 16
_precision = ["fp32"]


def compute_precision():
    return _precision[0]


# In the 16-bit operand mode the leaky-relu maps that only the WIDE 16-bit kernels read -- the transition's skip2, out3,
# out4, the encoder's first two hidden maps -- are STORED as IEEE half (scm_conv3x3_desc.y_half / x_half): the next layer
# and the weight gradient read two bytes per element, the value is the fp16 number the MFMA would have been fed anyway
# (forward bitwise unchanged), the activations saved for the backward pass take half the memory.  Gradients stay fp32 and
# travel beside the activations in cells (autograd would cast a gradient to its tensor's dtype): models.Transition /
# Encoder decide per call (half_store_ok).  Measured on one MI355X: the 128 -> 128 forward moves 184 MB
# instead of 353 MB per launch and takes 104 instead of 120 us; the maps that feed the 16-channel layers (skip1, out5,
# the encoder's third map) stay fp32 -- those kernels stage one element per lane and channel and were 40-55 % SLOWER on
# 2-byte loads.  +1.2 % (64x64) / +2.3 % (128x128x4, h = 20) end to end.  SCM_NO_HALF_STORE=1 / set_half_store(False):
# fp32 tensors everywhere.
use_half_store = os.environ.get("SCM_NO_HALF_STORE", "0") in ("", "0")


def set_half_store(on):
    global use_half_store
    use_half_store = bool(on)


def half_store_ok():
    return use_half_store and _precision[0] == "fp16" and use_gate_mask


def set_compute_precision(name):
    if name not in ("fp32", "fp16", "fp32x3"):
        raise ValueError("compute precision must be 'fp32', 'fp32x3' or 'fp16', got %r" % (name,))
    _precision[0] = name


# the environment default lets the whole fp32 parity suite run in another mode; validated like any other value
set_compute_precision(os.environ.get("SCM_PRECISION", "") or "fp32")


# fp32 convolution algorithm of the wide layers: "winograd" (default; F(2x2, 3x3), 4/9 of the direct form's
# products, fp32 operands and accumulation) or "direct" (implicit GEMM, scm_conv3x3).  Layers thinner than
# WINO_MIN_CH channels on either side, and frames scm_conv3x3_wino_serves() declines, run the direct kernels.
_conv_algo = [os.environ.get("SCM_CONV_ALGO", "") or "winograd"]
WINO_MIN_COUT = int(os.environ.get("SCM_WINO_MIN_COUT", "") or 64)
WINO_MIN_CIN = int(os.environ.get("SCM_WINO_MIN_CIN", "") or 8)


def conv_algo():
    return _conv_algo[0]


def set_conv_algo(name):
    if name not in ("winograd", "direct"):
        raise ValueError("convolution algorithm must be 'winograd' or 'direct', got %r" % (name,))
    _conv_algo[0] = name


set_conv_algo(_conv_algo[0])


WINO16 = os.environ.get("SCM_NO_WINO16", "0") in ("", "0")


def wino16_serves(H, W):
    return bool(L.lib().scm_conv3x3_wino16_serves(H, W))


def pack_mode():
    """What a cached weight pack depends on besides the weights: (operand precision, fp32 algorithm)."""
    return (_precision[0], _conv_algo[0])


def wino_serves(H, W):
    return bool(L.lib().scm_conv3x3_wino_serves(H, W))


# ------------------------------------------------------------------ packs
class PackSet(object):
    """(forward pack, dgrad pack, forward algorithm, dgrad algorithm, native forward pack, native dgrad pack) of a layer in
    the Winograd mode, indexable like the tuple it replaces.  The native packs of the directions the Winograd kernels
    serve are only a per-launch fallback: they are built when first asked for (items 4 / 5)."""

    def __init__(self, fpack, dpack, fprec, dprec, nfp, ndp, build):
        self._items = [fpack, dpack, fprec, dprec, nfp, ndp]
        self._build = build

    def _get(self, i):
        if i in (4, 5) and self._items[i] is None:
            f, d = self._build(i == 4, i == 5)
            self._items[i] = f if i == 4 else d
        return self._items[i]

    def __len__(self):
        return 6

    def __getitem__(self, key):
        if isinstance(key, slice):
            return tuple(self._get(i) for i in range(*key.indices(6)))
        return self._get(key if key >= 0 else key + 6)

    def __iter__(self):
        return (self._get(i) for i in range(6))


def pack_conv3x3(weight, cin, cout, transposed=False, cin_total=0, ci0=0):
    """Weight packs for scm_conv3x3 / scm_conv3x3_h16 from a torch-layout weight, in the
    current compute precision: (forward pack, dgrad pack, forward prec, dgrad prec)."""
    lib = L.lib()
    w = _chk(weight.detach(), "pack_conv3x3")
    if _precision[0] == "fp16":
        fp = torch.empty(lib.scm_conv3x3_pack_h16_bytes(cin, cout) // 2, device=w.device, dtype=torch.int16)
        dp = torch.empty(lib.scm_conv3x3_pack_h16_bytes(cout, cin) // 2, device=w.device, dtype=torch.int16)
        L.check(lib.scm_pack_conv3x3_weights_h16(_p(w), cin, cout, int(transposed), cin_total, ci0, _p(fp), PREC_F16,
                                                 _p(dp), PREC_BF16, _stream()), "scm_pack_conv3x3_weights_h16")
        return fp, dp, PREC_F16, PREC_BF16
    def native(want_f, want_d):
        f = _new(w, lib.scm_conv3x3_pack_floats(cin, cout)) if want_f else None
        d = _new(w, lib.scm_conv3x3_pack_floats(cout, cin)) if want_d else None
        L.check(lib.scm_pack_conv3x3_weights(_p(w), cin, cout, int(transposed), cin_total, ci0, _p(f), _p(d), _stream()),
                "scm_pack_conv3x3_weights")
        return f, d
    if _precision[0] == "fp32" and _conv_algo[0] == "winograd":
        # transformed-filter packs for the directions wide enough on both sides; the native packs of those directions are
        # only the per-call fallback for frames / operand layouts the kernels do not take (wino_serves): built on first use
        # (<= 16 output channels: the 16-row form, scm_conv3x3_wino16_serves; from 32 input channels on)
        fwd_w = (cout >= WINO_MIN_COUT and cin >= WINO_MIN_CIN) or (cout <= 16 and cin >= 32 and WINO16)
        dgr_w = (cin >= WINO_MIN_COUT and cout >= WINO_MIN_CIN) or (cin <= 16 and cout >= 32 and WINO16)
        if fwd_w or dgr_w:
            fx = _new(w, lib.scm_conv3x3_pack_wino_floats(cin, cout)) if fwd_w else None
            dx = _new(w, lib.scm_conv3x3_pack_wino_floats(cout, cin)) if dgr_w else None
            L.check(lib.scm_pack_conv3x3_weights_wino(_p(w), cin, cout, int(transposed), cin_total, ci0, _p(fx), _p(dx),
                                                      _stream()), "scm_pack_conv3x3_weights_wino")
            fp, dp = native(not fwd_w, not dgr_w) if not (fwd_w and dgr_w) else (None, None)
            return PackSet(fx if fwd_w else fp, dx if dgr_w else dp, ALGO_WINO if fwd_w else PREC_F32,
                           ALGO_WINO if dgr_w else PREC_F32, fp, dp, native)
    fp, dp = native(True, True)
    if _precision[0] == "fp32x3":
        # split packs for the directions the x3 kernel serves (> 16 output channels); the native packs stay as the
        # per-call fallback for frames it does not take (x3_serves)
        # measured per direction: the split forward wins from 16 input channels on (67.3 vs 66.0 ms per step with the
        # 16 -> 128 layers native); the input-gradient conv of a 16-channel output gradient (conv6: one K chunk into 256
        # channels, 255 vs 207 us) does not
        fx = dx = None
        if cout > 16:
            fx = torch.empty(lib.scm_conv3x3_pack_x3_bytes(cin, cout) // 2, device=w.device, dtype=torch.int16)
        if cin > 16 and cout >= 32:
            dx = torch.empty(lib.scm_conv3x3_pack_x3_bytes(cout, cin) // 2, device=w.device, dtype=torch.int16)
        if fx is not None or dx is not None:
            L.check(lib.scm_pack_conv3x3_weights_x3(_p(w), cin, cout, int(transposed), cin_total, ci0, _p(fx), _p(dx),
                                                    _stream()), "scm_pack_conv3x3_weights_x3")
        return (fx if fx is not None else fp, dx if dx is not None else dp, PREC_BF16X3 if fx is not None else PREC_F32,
                PREC_BF16X3 if dx is not None else PREC_F32, fp, dp)
    return fp, dp, PREC_F32, PREC_F32


def _halo_plane(H, W, NP):
    rows = NP // W if NP % W == 0 else (NP + W - 2) // W + 1
    return (min(rows, H) + 2) * (W + 2)


def x3_serves(H, W, B=None, cout=None):
    """Whether scm_conv3x3_x3 takes this launch: its 512-pixel workgroups stage the halo tile in 768 slots per channel.
    With a batch size: a launch that fills less than half of the 256 CUs with 512-pixel workgroups (configs[0], the
    planner's beams, the eval rollout) is taken only where the kernel has its 128-pixel form (more than 64 output
    channels, halo tile within 512 slots); thinner layers then stay on the native kernel's small tiles."""
    if _halo_plane(H, W, 512) > 768:
        return False
    if B is None or os.environ.get("SCM_X3_ANY_BATCH"):      # (the parity suite runs its small batches through the split kernels)
        return True
    groups = 1 if cout is None else (cout + 127) // 128
    if B * ((H * W + 511) // 512) * groups >= 128:
        return True
    return cout is not None and cout > 64 and _halo_plane(H, W, 128) <= 512


class KernelTimer(object):
    """Optional HIP-event bracket around kernel launches (bench.py roofline):
    events are recorded on the stream the kernels are launched on."""

    def __init__(self):
        self.records = []

    def bracket(self, tag, flops):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.records.append((tag, flops, s, e))
        return s, e

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for tag, flops, s, e in self.records:
            t = out.setdefault(tag, {"launches": 0, "ms": 0.0, "flops": 0.0})
            t["launches"] += 1
            t["ms"] += s.elapsed_time(e)
            t["flops"] += flops
        return out


kernel_timer = None     # set to a KernelTimer() to time the MFMA conv launches


def conv_kernel_name(cout, prec=0, cin=None):
    if prec == ALGO_WINO:
        # (the entry point sends layers with at most 16 output / input channels to the 16-row / 16-input-channel kernels:
        # their own buckets, so that the bracketed average of the wide kernel is that kernel's and agrees with rocprofv3's)
        if cout <= 16:
            return "conv3x3_wino16_kernel"
        return "conv3x3_wino_k16_kernel" if (cin is not None and cin <= 16) else "conv3x3_wino_kernel"
    if prec == PREC_BF16X3:
        return "conv3x3_x3_kernel"
    if prec:
        return "conv3x3_h16_kernel<%s>" % ("f16" if prec == PREC_F16 else "bf16")
    if cout <= 16:
        return "conv3x3_m16_kernel"
    return "conv3x3_mfma_kernel<%s>" % ("4,2|4,4" if cout > 64 else ("2,4" if cout > 32 else "1,8"))


# The leaky-relu gate of the gradient kernels travels as a 1-bit mask written by the forward
# epilogue (scm_conv3x3_desc.gate_mask_out): a gradient convolution then loads one word per pixel
# and 32 channels instead of 32 floats.  Layers with cout <= 32 keep the float gate only.
use_gate_mask = os.environ.get("SCM_NO_GATE_MASK", "0") in ("", "0")


def _conv3x3_raw(x1, x2, gate1, gate2, gate_act, wpack, cout, scale, bias, sbias, co1, pad, act, prec=0,
                 mask_out=None, mask_in=None, y1_acc=None, ymask1=None, ymask2=None, y_half=False):
    """y1_acc: an existing (B, co1, H, W) tensor the first co1 output channels are ADDED to (accumulate_y1).
    ymask1 / ymask2: leaky-relu gate masks of the activations whose gradients y1 / y2 are (gate_mask_y1 / _y2:
    the Winograd kernel multiplies what it writes by the slope of its own element)."""
    lib = L.lib()
    B, c1, H, W = x1.shape
    c2 = 0 if x2 is None else x2.shape[1]
    x_half = x1.dtype == torch.float16
    if x_half or y_half:
        if prec != PREC_F16 or (x2 is not None and x2.dtype != x1.dtype) or (y_half and y1_acc is not None):
            raise RuntimeError("scm_gan_amd.conv3x3: half-stored activations are read / written by the fp16 forward "
                               "convolution only")
    ydt = torch.float16 if y_half else torch.float32
    y1 = torch.empty((B, co1, H, W), device=x1.device, dtype=ydt) if y1_acc is None else y1_acc
    y2 = torch.empty((B, cout - co1, H, W), device=x1.device, dtype=ydt) if co1 < cout else None
    d = L.ConvDesc(_p(x1), _p(x2), c1, c2, _p(gate1), _p(gate2), gate_act, _p(wpack), cout, _p(scale), _p(bias),
                   _p(sbias), _p(y1), _p(y2), co1, B, H, W, pad, act, _p(mask_out), _p(mask_in),
                   0 if y1_acc is None else 1, _p(ymask1), _p(ymask2), int(x_half), int(y_half))
    if kernel_timer is not None:
        ev = kernel_timer.bracket(conv_kernel_name(cout, prec, c1 + c2), 2.0 * B * H * W * cout * (c1 + c2) * 9)
        ev[0].record()
    if prec == ALGO_WINO:
        L.check(lib.scm_conv3x3_wino(ctypes.byref(d), _stream()), "scm_conv3x3_wino")
    elif prec == PREC_BF16X3:
        L.check(lib.scm_conv3x3_x3(ctypes.byref(d), _stream()), "scm_conv3x3_x3")
    elif prec:
        L.check(lib.scm_conv3x3_h16(ctypes.byref(d), prec, _stream()), "scm_conv3x3_h16")
    else:
        L.check(lib.scm_conv3x3(ctypes.byref(d), _stream()), "scm_conv3x3")
    if kernel_timer is not None:
        ev[1].record()
    return y1, y2


def sn_power_iter(weight_bar, u, v):
    """One SpectralNorm._update_u_v call (spectral_normalization.py:23-35):
    updates u, v in place, returns (sigma, inv_sigma) as 1-element tensors."""
    lib = L.lib()
    w = _chk(weight_bar.detach(), "sn_power_iter")
    rows = w.shape[0]
    cols = w.numel() // rows
    out = _new(w, 2)
    sigma, inv_sigma = out[0:1], out[1:2]
    layer = L.SnLayer(_p(w), _p(u.data), _p(v.data), _p(sigma), _p(inv_sigma), rows, cols)
    L.check(lib.scm_sn_power_iter(1, ctypes.byref(layer), 1, _stream()), "scm_sn_power_iter")
    return sigma, inv_sigma


def sn_power_iter_multi(layers, steps):
    """`steps` consecutive power iterations for up to 8 (weight_bar, u, v) triples in
    ONE launch (one workgroup per layer).  Returns a (nlayers, 2, steps) tensor:
    [:, 0] = sigma after each call, [:, 1] = 1/sigma."""
    lib = L.lib()
    n = len(layers)
    w0 = layers[0][0]
    out = _new(w0, n, 2, steps)
    arr = (L.SnLayer * n)()
    for i, (w, u, v) in enumerate(layers):
        w = _chk(w.detach(), "sn_power_iter_multi")
        rows = w.shape[0]
        arr[i] = L.SnLayer(_p(w), _p(u.data), _p(v.data), _p(out[i, 0]), _p(out[i, 1]), rows, w.numel() // rows)
    L.check(lib.scm_sn_power_iter(n, arr, steps, _stream()), "scm_sn_power_iter")
    return out


def _sn_fold(g, weight_bar, u, v, inv_sigma, dot_part=None, into=None):
    """dW_bar from dW_eff; `into` = an existing gradient buffer to accumulate into."""
    lib = L.lib()
    rows = weight_bar.shape[0]
    cols = weight_bar.numel() // rows
    out = torch.empty_like(g) if into is None else into
    ndot = 0 if dot_part is None else dot_part.numel()
    L.check(lib.scm_sn_fold_grad(_p(g), _p(weight_bar), _p(u), _p(v), _p(inv_sigma), _p(dot_part), ndot, rows, cols,
                                 0 if into is None else 1, _p(out), _stream()), "scm_sn_fold_grad")
    return out


# Inside `with grad_sinks():` (Trainer.step's backward), a parameter that already owns a
# contiguous fp32 .grad buffer (the flat buckets) gets its weight / bias gradients accumulated
# into that buffer by the backward kernels themselves, and autograd is handed no gradient for
# it: saves one elementwise add launch per parameter per timestep.  Off by default, so
# torch.autograd.grad(), backward(inputs=...) and parameter hooks keep their meaning.
_sinks_on = [False]


class grad_sinks(object):
    def __init__(self, on=True):
        self.on = on

    def __enter__(self):
        self.prev = _sinks_on[0]
        _sinks_on[0] = self.on

    def __exit__(self, *exc):
        _sinks_on[0] = self.prev


def _grad_sink(param):
    if not _sinks_on[0] or param is None or not isinstance(param, torch.nn.Parameter):
        return None
    g = param.grad
    if g is None or not g.is_contiguous() or g.dtype != torch.float32 or not g.is_cuda:
        return None
    return g


# Producer-side gating of gradients: inside Encoder / Transition every consumer of a leaky-relu
# activation is a wide convolution whose input-gradient kernel (Winograd) can multiply what it writes by the slope of
# the element it writes to -- the layer that produced the activation then receives a gradient that is already gated
# and runs its own gradient kernels without a gate.  The models decide per call (pregate_ok); SCM_NO_PREGATE=1 keeps
# the consumer-side gate everywhere.
use_pregate = os.environ.get("SCM_NO_PREGATE", "0") in ("", "0")


def pregate_ok(H, W):
    """Whether the gradients of a leaky-relu activation map of this frame size can be gated by their producers: the
    default fp32 Winograd kernels must serve every input-gradient convolution of the chain."""
    return (use_pregate and use_gate_mask and _conv_algo[0] == "winograd" and _precision[0] == "fp32"
            and WINO_MIN_CIN <= 16 and WINO_MIN_COUT <= 64 and wino_serves(H, W))


def _gate_by_mask(dx, mask):
    """dx * leaky-relu slope of the activation whose 1-bit gate mask (B, ceil(C/32), H, W) this is."""
    if dx is None or mask is None:
        return dx
    C = dx.shape[1]
    ch = torch.arange(C, device=dx.device)
    bits = (mask[:, ch // 32] >> (ch % 32).to(torch.int32).view(1, C, 1, 1)) & 1
    return dx * torch.where(bits.bool(), 1.0, 0.01).to(dx.dtype)      # models.py: leaky_relu default slope


# -------------------------------------------------------------- conv 3x3
class Conv3x3Fn(Function):
    """y = act(conv3x3(cat(x1, x2), W_eff) + bias [+ action-plane bias]),
    W_eff = weight * inv_sigma (SpectralNorm) or weight.

    cfg: pad, act, transposed (ConvTranspose2d weight layout), packs (fwd, dgrad),
         cin_conv (input channels convolved; the remaining weight channels are
         the action planes folded into a per-sample bias).
    """

    @staticmethod
    def forward(ctx, x1, x2, act_vec, weight, bias, inv_sigma, u, v, cfg):
        lib = L.lib()
        x1 = _chk(x1, "conv3x3", half_ok=True)
        x2 = _chk(x2, "conv3x3", half_ok=True)
        act_vec = _chk(act_vec, "conv3x3")
        w = _chk(weight.detach(), "conv3x3")
        B, c1, H, W = x1.shape
        c2 = 0 if x2 is None else x2.shape[1]
        if cfg.get("out_cell") is not None:
            # this layer's output is a half tensor: its fp32 gradient arrives through the cell, not through autograd
            # (which would cast it to half); the consumers return None for it
            ctx.set_materialize_grads(False)
        cout = w.shape[1] if cfg["transposed"] else w.shape[0]
        cin_total = w.shape[0] if cfg["transposed"] else w.shape[1]
        fpack, dpack, fprec, dprec = cfg["packs"][:4]
        if fprec == PREC_BF16X3 and not x3_serves(H, W, B, cout):
            fpack, fprec = cfg["packs"][4], PREC_F32
        if dprec == PREC_BF16X3 and not x3_serves(H, W, B, c1 + c2):
            dpack, dprec = cfg["packs"][5], PREC_F32
        if ALGO_WINO in (fprec, dprec):
            # frames / operand layouts the Winograd kernels do not take run the direct kernels (native packs)
            frame_ok = wino_serves(H, W)
            f_ok = wino16_serves(H, W) if cout <= 16 else frame_ok
            if fprec == ALGO_WINO and not (f_ok and (c2 == 0 or c1 % 8 == 0) and x1.data_ptr() % 8 == 0
                                           and (x2 is None or x2.data_ptr() % 8 == 0)):
                fpack, fprec = cfg["packs"][4], PREC_F32
            if dprec == ALGO_WINO and not (wino16_serves(H, W) if c1 + c2 <= 16 else frame_ok):
                dpack, dprec = cfg["packs"][5], PREC_F32
        sbias = None
        if act_vec is not None:
            A = act_vec.shape[1]
            assert c1 + c2 + A == cin_total
            sbias = _new(x1, B, cout)
            L.check(lib.scm_action_bias(_p(act_vec), _p(w), _p(inv_sigma), B, A, cout, cin_total, c1 + c2, _p(sbias),
                                        _stream()), "scm_action_bias")
        mask = None
        if use_gate_mask and cfg["act"] == ACT_LRELU and cout > 32:
            mask = torch.empty((B, (cout + 31) // 32, H, W), device=x1.device, dtype=torch.int32)
        y, _ = _conv3x3_raw(x1, x2, None, None, 0, fpack, cout, inv_sigma, bias, sbias, cout, cfg["pad"], cfg["act"],
                            fprec, mask_out=mask, y_half=bool(cfg.get("out_half")))
        if cfg.get("out_half") and torch.is_grad_enabled() and (mask is None or cfg.get("out_cell") is None):
            raise RuntimeError("scm_gan_amd.conv3x3: a half-stored output needs its 1-bit gate and a gradient cell")
        if cfg.get("mask_holder") is not None:
            cfg["mask_holder"]["mask"] = mask          # for the consumers of y (producer-side gating)
        ctx.cfg = cfg
        ctx.dprec = dprec
        ctx.mask = mask
        ctx.shape = (B, c1, c2, H, W, cout, cin_total)
        ctx.has = (x2 is not None, act_vec is not None, bias is not None, inv_sigma is not None)
        ctx.params = (weight, bias)          # the Parameter objects: their .grad is looked up at backward time
        ctx.save_for_backward(x1, x2, act_vec, w, inv_sigma, u, v, y, dpack)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = L.lib()
        x1, x2, act_vec, w, inv_sigma, u, v, y, dpack = ctx.saved_tensors
        cfg = ctx.cfg
        B, c1, c2, H, W, cout, cin_total = ctx.shape
        if cfg.get("out_cell") is not None:
            gy = cfg["out_cell"].pop("buf", None)        # (left there by the consumers' backward passes)
            if gy is None:
                raise RuntimeError("scm_gan_amd.conv3x3.backward: no gradient was left in the cell of a half-stored "
                                   "activation (its consumers must be convolutions given dx1_cell / dx2_cell)")
        gy = _chk(gy, "conv3x3.backward")
        half_y = y.dtype == torch.float16
        gate_t = gy if half_y else y        # (a half y is never read as a gate: its 1-bit mask is; the pointer must be valid)
        half_x1, half_x2 = x1.dtype == torch.float16, (x2 is not None and x2.dtype == torch.float16)
        act = cfg["act"]
        gmask = ctx.mask
        if cfg.get("pregated"):
            # every producer of gy multiplied it by this layer's leaky-relu slopes already (gate_mask_y1 / _y2)
            assert act == ACT_LRELU and ctx.mask is not None
            act, gmask = ACT_NONE, None
        in_masks = cfg.get("in_masks") or (None, None)
        need_x1, need_x2, need_a, need_w, need_b = ctx.needs_input_grad[:5]
        dx1 = dx2 = d_act = dw = db = None
        if need_x1 or need_x2:
            # an input with a second consumer (SharedGradFork): whichever consumer's backward runs first leaves its
            # input gradient in the shared cell, the other adds into that buffer in its gradient conv's epilogue
            cell1, cell2 = cfg.get("dx1_cell"), cfg.get("dx2_cell")
            acc1 = cell1.get("buf") if cell1 is not None else None
            dprec = ctx.dprec
            if dprec == ALGO_WINO and (gy.data_ptr() | y.data_ptr()) % 8:
                dpack, dprec = cfg["packs"][5], PREC_F32          # (an oddly offset gradient view: direct kernel)
            if dprec == ALGO_WINO and c1 + c2 <= 16 and (act or acc1 is not None or c2):
                dpack, dprec = cfg["packs"][5], PREC_F32          # (the 16-row form takes ungated, plain-store launches)
            if (in_masks[0] is not None or in_masks[1] is not None) and dprec != ALGO_WINO:
                # The chain was forwarded with producer-side gating (ops.pregate_ok looks at the frame), but THIS layer's
                # input gradient left the Winograd kernel (a latent_size below its channel minimum, an oddly offset
                # gradient view): the direct kernel writes the ungated gradient and the destination's slopes are applied
                # from its 1-bit mask here, before anything is added to a shared buffer -- rare, correct, not fast.
                dx1, dx2 = _conv3x3_raw(gy, None, gate_t if act else None, None, act, dpack, c1 + c2, inv_sigma, None, None,
                                        c1, cfg["pad"], ACT_NONE, dprec, mask_in=gmask)
                dx1, dx2 = _gate_by_mask(dx1, in_masks[0]), _gate_by_mask(dx2, in_masks[1])
                if acc1 is not None:
                    dx1 = acc1.add_(dx1)
            else:
                dx1, dx2 = _conv3x3_raw(gy, None, gate_t if act else None, None, act, dpack, c1 + c2, inv_sigma, None, None, c1,
                                        cfg["pad"], ACT_NONE, dprec, mask_in=gmask, y1_acc=acc1, ymask1=in_masks[0],
                                        ymask2=in_masks[1])
            if (half_x1 and cell1 is None) or (half_x2 and cell2 is None):
                raise RuntimeError("scm_gan_amd.conv3x3.backward: a half-stored input needs a gradient cell")
            if cell1 is not None:
                cell1["buf"] = dx1
            if cell2 is not None and dx2 is not None:
                if cell2.get("buf") is None:
                    cell2["buf"] = dx2
                else:                       # (not the order autograd produces for the transition; kept correct)
                    cell2["buf"].add_(dx2)
                    dx2 = cell2["buf"]
        if need_w or need_b or need_a:
            pf, cf = ctypes.c_size_t(), ctypes.c_size_t()
            L.check(lib.scm_conv3x3_wgrad_workspace(B, H, W, c1 + c2, cout, ctypes.byref(pf), ctypes.byref(cf)),
                    "scm_conv3x3_wgrad_workspace")
            part = _new(gy, pf.value)
            cws = _new(gy, cf.value)
            has_act = act_vec is not None
            w_sink, b_sink = _grad_sink(ctx.params[0]), _grad_sink(ctx.params[1])
            direct_w = w_sink is not None and inv_sigma is None and not has_act    # reduce straight into .grad
            # dw_eff: gradient w.r.t. the effective weight, in the parameter's layout
            dw_eff = w_sink if direct_w else torch.empty_like(w)
            db = b_sink if b_sink is not None else _new(gy, cout)
            acc_flag = (1 if direct_w else 0) | (2 if b_sink is not None else 0)
            csum = _new(gy, B, cout) if has_act else None
            # <dW_eff, W_bar> partials for the SpectralNorm fold come out of the reduce kernel
            # (not when action channels add their slice afterwards: the fold then does the dot itself)
            dot_part = None
            if inv_sigma is not None and not has_act:
                dot_part = _new(gy, (9 * cout * (c1 + c2) + 255) // 256)
            d = L.WgradDesc(_p(x1), _p(x2), c1, c2, _p(gy), _p(gate_t) if act else None, act, cout, _p(dw_eff), _p(db),
                            _p(csum), _p(part), _p(cws), B, H, W, cfg["pad"], int(cfg["transposed"]), acc_flag,
                            cin_total, 0, _p(w) if dot_part is not None else None, _p(dot_part), _p(gmask),
                            int(half_x1))
            wprec = ctx.dprec if ctx.dprec in (PREC_F16, PREC_BF16) else PREC_F32
            if compute_precision() == "fp32x3":
                # the split kernel takes 128-wide weight gradients of layers with more than 16 input channels (at any batch:
                # its work split fills the chip also where the forward's 512-pixel tiles do not)
                wprec = PREC_BF16X3 if (W >= 8 and x3_serves(H, W) and cout > 64 and c1 + c2 > 16) else PREC_F32
            thin16 = (cout <= 16 or c1 + c2 <= 16) and not half_x1 and not half_y and _conv_algo[0] == "winograd"
            if wprec in (PREC_F16, PREC_BF16) and thin16 and gmask is not None:
                d.gate_mask = None          # (the thin Winograd form takes the float gate: y is an fp32 tensor here)
            if wprec in (PREC_F16, PREC_BF16) and thin16:
                # 16-bit mode, a layer with at most 16 channels on one side (Transition.conv6 / conv1, Encoder.conv4 /
                # conv1): the 16-bit weight-gradient kernel does not take it and the direct fp32 kernel it falls back to is
                # slower than the fp32 Winograd form of the same gradient (conv6: 184 vs 130 us) -- also the more accurate one
                L.check(lib.scm_conv3x3_wgrad_wino(ctypes.byref(d), _stream()), "scm_conv3x3_wgrad_wino")
            elif wprec:
                L.check(lib.scm_conv3x3_wgrad_h16(ctypes.byref(d), wprec, _stream()), "scm_conv3x3_wgrad_h16")
            elif _conv_algo[0] == "winograd" and compute_precision() == "fp32":
                # (takes the wide layers, forwards the rest to the direct kernels itself)
                L.check(lib.scm_conv3x3_wgrad_wino(ctypes.byref(d), _stream()), "scm_conv3x3_wgrad_wino")
            else:
                L.check(lib.scm_conv3x3_wgrad(ctypes.byref(d), _stream()), "scm_conv3x3_wgrad")
            if has_act:
                A = act_vec.shape[1]
                d_act = _new(gy, B, A)
                L.check(lib.scm_action_bias_bwd(_p(act_vec), _p(w), _p(inv_sigma), _p(csum), B, A, cout, cin_total,
                                                c1 + c2, _p(d_act), _p(dw_eff), _stream()), "scm_action_bias_bwd")
            if inv_sigma is not None:
                dw = _sn_fold(dw_eff, w, u, v, inv_sigma, dot_part, into=w_sink)
                sunk_w = w_sink is not None
            else:
                dw = dw_eff
                sunk_w = direct_w
            if sunk_w:
                dw = None                # already accumulated into weight.grad
            if b_sink is not None:
                db = None
        # (the gradient of a half-stored input stays in its cell: its producer's backward takes it from there)
        return (dx1 if (need_x1 and not half_x1) else None, dx2 if (need_x2 and not half_x2) else None,
                d_act if need_a else None, dw if need_w else None, db if need_b else None, None, None, None, None)


class SharedGradFork(Function):
    """x -> (x, x) for an activation with two convolution consumers.  `cell` is the dict the two consumers were
    given as dx1_cell / dx2_cell: their backward passes write / accumulate the input gradient into ONE buffer
    (scm_conv3x3_desc.accumulate_y1), so the sum autograd would compute with an elementwise add kernel (a 128-channel
    map read twice and written once) never runs.  Falls back to that add when the gradients arrive separately."""

    @staticmethod
    def forward(ctx, x, cell):
        ctx.cell = cell
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, g1, g2):
        ctx.cell.pop("buf", None)
        if g1 is None or g2 is None:
            return (g1 if g2 is None else g2), None
        if g1.data_ptr() == g2.data_ptr():
            return g1, None
        return g1 + g2, None


# ------------------------------------------------------------ small conv
class Conv2dFn(Function):
    """Generic small Conv2d (K, stride, pad) + activation; optional SpectralNorm."""

    @staticmethod
    def forward(ctx, x, weight, bias, inv_sigma, u, v, K, stride, pad, act):
        lib = L.lib()
        x = _chk(x, "conv2d")
        w = _chk(weight.detach(), "conv2d")
        B, cin, H, W = x.shape
        cout = w.shape[0]
        OH, OW = (H + 2 * pad - K) // stride + 1, (W + 2 * pad - K) // stride + 1
        y = _new(x, B, cout, OH, OW)
        d = L.Conv2dDesc(_p(x), _p(w), _p(bias), _p(inv_sigma), _p(y), None, 0, B, cin, H, W, cout, K, stride, pad, act)
        L.check(lib.scm_conv2d_fwd(ctypes.byref(d), _stream()), "scm_conv2d_fwd")
        ctx.geom = (B, cin, H, W, cout, K, stride, pad, act)
        ctx.save_for_backward(x, w, inv_sigma, u, v, y)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = L.lib()
        x, w, inv_sigma, u, v, y = ctx.saved_tensors
        B, cin, H, W, cout, K, stride, pad, act = ctx.geom
        gy = _chk(gy, "conv2d.backward")
        need_x, need_w, need_b = ctx.needs_input_grad[:3]
        dx = torch.empty_like(x) if need_x else None
        dw_eff = torch.empty_like(w) if (need_w or need_b) else None
        db = _new(gy, cout) if (need_w or need_b) else None
        d = L.Conv2dDesc(_p(x), _p(w), None, _p(inv_sigma), None, _p(y) if act else None, act, B, cin, H, W, cout, K,
                         stride, pad, act)
        ws = _new(gy, lib.scm_conv2d_bwd_workspace(B, cin, cout, K)) if dw_eff is not None else None
        L.check(lib.scm_conv2d_bwd(ctypes.byref(d), _p(gy), _p(dx), _p(dw_eff), _p(db), 0, _p(ws), _stream()),
                "scm_conv2d_bwd")
        dw = dw_eff
        if dw_eff is not None and inv_sigma is not None:
            dw = _sn_fold(dw_eff, w, u, v, inv_sigma)
        return dx, dw if need_w else None, db if need_b else None, None, None, None, None, None, None, None


# ----------------------------------------------------------- elementwise
class BernoulliSTFn(Function):
    """DifferentiableBernoulliSampler (models.py:29-40): forward samples
    (noise < p), backward passes the gradient through unchanged."""

    @staticmethod
    def forward(ctx, p, noise, mode, seed, offset):
        p = _chk(p, "bernoulli")
        noise = _chk(noise, "bernoulli")
        out = torch.empty_like(p)
        L.check(L.lib().scm_bernoulli(_p(p), _p(noise), _p(out), p.numel(), mode, seed, offset,
                                      _dev("noise_off") if mode == 2 else None, _stream()), "scm_bernoulli")
        return out

    @staticmethod
    def backward(ctx, g):
        return g, None, None, None, None


def threshold(p):
    """Eval-mode latent: (p > 0.5), models.py:110-113 (no gradient)."""
    p = _chk(p.detach(), "threshold")
    out = torch.empty_like(p)
    L.check(L.lib().scm_bernoulli(_p(p), None, _p(out), p.numel(), 0, 0, 0, None, _stream()), "scm_bernoulli")
    return out


def bernoulli_sample(p, seed, offset):
    """(philox(seed, offset) < p) without a gradient: the U{0,1} noise maps the discriminator
    is fed (models.py:18,193-199) are bernoulli_sample(0.5)."""
    p = _chk(p.detach(), "bernoulli_sample")
    out = torch.empty_like(p)
    L.check(L.lib().scm_bernoulli(_p(p), None, _p(out), p.numel(), 2, int(seed), int(offset), _dev("noise_off"),
                                  _stream()), "scm_bernoulli")
    return out


class FactorSumFn(Function):
    """x.view(B, L, C, H, W).sum(1), models.py:280-287."""

    @staticmethod
    def forward(ctx, x, L_, C):
        x = _chk(x, "factor_sum")
        B, _, H, W = x.shape
        y = _new(x, B, C, H, W)
        L.check(L.lib().scm_factor_sum(_p(x), _p(y), B, L_, C, H * W, _stream()), "scm_factor_sum")
        ctx.geom = (B, L_, C, H, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        B, L_, C, H, W = ctx.geom
        gy = _chk(gy, "factor_sum.backward")
        dx = _new(gy, B, L_ * C, H, W)
        L.check(L.lib().scm_factor_bcast(_p(gy), _p(dx), B, L_, C, H * W, _stream()), "scm_factor_bcast")
        return dx, None, None


class PixelLossFn(Function):
    """per_sample[b] = mean_{chw} loss(sigmoid(logit), target); kind 0 = BCE
    (main.py:189-190,310-312), kind 1 = squared error (main.py:814-816)."""

    @staticmethod
    def forward(ctx, logit, target, kind):
        logit = _chk(logit, "pixel_loss")
        target = _chk(target, "pixel_loss")
        B = logit.shape[0]
        n = logit.numel() // B
        out = _new(logit, B)
        L.check(L.lib().scm_pixel_loss(_p(logit), _p(target), B, n, kind, _p(out), None, _stream()), "scm_pixel_loss")
        ctx.kind = kind
        ctx.save_for_backward(logit, target)
        return out

    @staticmethod
    def backward(ctx, g):
        logit, target = ctx.saved_tensors
        g = _chk(g, "pixel_loss.backward")
        B = logit.shape[0]
        n = logit.numel() // B
        d = torch.empty_like(logit)
        L.check(L.lib().scm_pixel_loss_bwd(_p(logit), _p(target), _p(g), B, n, ctx.kind, _p(d), _stream()),
                "scm_pixel_loss_bwd")
        return d, None, None


class RewardTailFn(Function):
    """softmax over 3 classes -> p0 - p2 -> (sum over h,w ; map), models.py:241-250."""

    @staticmethod
    def forward(ctx, x, R):
        x = _chk(x, "reward_tail")
        B, _, h, w = x.shape
        r = _new(x, B, R)
        m = _new(x, B, R, h, w)
        L.check(L.lib().scm_reward_tail(_p(x), B, R, h * w, _p(r), _p(m), _stream()), "scm_reward_tail")
        ctx.geom = (B, R, h, w)
        ctx.save_for_backward(x)
        ctx.set_materialize_grads(False)        # an unused output (the map, usually) contributes no zero-filled gradient
        return r, m

    @staticmethod
    def backward(ctx, g_r, g_map):
        (x,) = ctx.saved_tensors
        B, R, h, w = ctx.geom
        if g_r is None and g_map is None:
            return None, None
        g_r = _chk(g_r, "reward_tail.backward")
        g_map = _chk(g_map, "reward_tail.backward")
        dx = torch.empty_like(x)
        L.check(L.lib().scm_reward_tail_bwd(_p(x), _p(g_r), _p(g_map), B, R, h * w, _p(dx), _stream()),
                "scm_reward_tail_bwd")
        return dx, None


class LinearFn(Function):
    """act(x w^T + b), models.py:205,221-222."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        x = _chk(x, "linear")
        w = _chk(weight.detach(), "linear")
        B, nin = x.shape
        nout = w.shape[0]
        y = _new(x, B, nout)
        L.check(L.lib().scm_linear(_p(x), _p(w), _p(bias), B, nin, nout, act, _p(y), _stream()), "scm_linear")
        ctx.act = act
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, w, y)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        gy = _chk(gy, "linear.backward")
        B, nin = x.shape
        nout = w.shape[0]
        dx, dw, db = torch.empty_like(x), torch.empty_like(w), _new(gy, nout)
        L.check(L.lib().scm_linear_bwd(_p(x), _p(w), _p(y), _p(gy), B, nin, nout, ctx.act, _p(dx), _p(dw), _p(db),
                                       _stream()), "scm_linear_bwd")
        return dx, dw, (db if ctx.has_bias else None), None


# ---------------------------------------------------------- fused decoder
def decoder_fused_pack(w1, b1, w2, b2, latent, C):
    """Weight packs of the fused decoder kernels from the four ConvTranspose2d parameters."""
    lib = L.lib()
    ts = [_chk(t.detach(), "decoder_fused_pack") for t in (w1, b1, w2, b2)]
    pack = _new(ts[0], lib.scm_decoder_fused_pack_floats())
    L.check(lib.scm_decoder_fused_pack(_p(ts[0]), _p(ts[1]), _p(ts[2]), _p(ts[3]), latent, C, _p(pack), _stream()),
            "scm_decoder_fused_pack")
    return pack


def _decoder_fused_ws(B, H, W, C):
    f, b = ctypes.c_size_t(), ctypes.c_size_t()
    L.check(L.lib().scm_decoder_fused_workspace(B, H, W, C, ctypes.byref(f), ctypes.byref(b)), "scm_decoder_fused_workspace")
    return f.value, b.value


# The fused heads keep their hidden maps (64 / 32 channels) for the backward pass, which then skips recomputing them:
# a quarter of its MFMAs.  SCM_NO_SAVED_HIDDEN=1: recompute from z (round 2's form; the A/B).
save_hidden = os.environ.get("SCM_NO_SAVED_HIDDEN", "0") in ("", "0")


def decoder_fused_fwd(z, pack, C, target=None, loss_kind=-1, want_logits=True, h_save=None):
    """(logits or None, per-sample loss or None) of the fused decoder forward (no autograd)."""
    z = _chk(z, "decoder_fused")
    target = _chk(target, "decoder_fused")
    B, _, H, W = z.shape
    logits = _new(z, B, C, H, W) if want_logits else None
    loss = _new(z, B) if loss_kind >= 0 else None
    ws = _new(z, max(1, _decoder_fused_ws(B, H, W, C)[0])) if loss_kind >= 0 else None
    d = L.DecoderFusedDesc(_p(z), _p(pack), _p(target), _p(logits), _p(loss), _p(ws), B, H, W, C, loss_kind, _p(h_save))
    if kernel_timer is not None:
        ev = kernel_timer.bracket("decoder_fused_fwd", 2.0 * B * H * W * 9 * (16 * 64 + 64 * C))
        ev[0].record()
    L.check(L.lib().scm_decoder_fused_fwd(ctypes.byref(d), _stream()), "scm_decoder_fused_fwd")
    if kernel_timer is not None:
        ev[1].record()
    return logits, loss


class DecoderLossFn(Function):
    """Per-sample pixel loss of the decoded frame in one pass over z (scm_decoder_fused_fwd / _bwd):
    loss[b] = mean_{c,h,w} loss(sigmoid(Decoder(z)), target), kind 0 = BCE (main.py:188-190, 310-312), kind 1 =
    squared error (main.py:814-816).  The 64-channel hidden map exists only inside the kernels; backward
    recomputes it from z.  Returns (loss, logits); logits carry no gradient."""

    @staticmethod
    def forward(ctx, z, target, w1, b1, w2, b2, pack, kind, keep_hidden=True):
        z = _chk(z, "decoder_loss")
        target = _chk(target, "decoder_loss")
        C = target.shape[1]
        # (keep_hidden: the caller's torch.is_grad_enabled() -- inside forward() grad mode is always off)
        hid = _new(z, z.shape[0], 64, z.shape[2], z.shape[3]) if (save_hidden and keep_hidden and any(ctx.needs_input_grad)) else None
        logits, loss = decoder_fused_fwd(z, pack, C, target, kind, True, hid)
        ctx.kind, ctx.C = kind, C
        ctx.params = (w1, b1, w2, b2)
        ctx.save_for_backward(z, target, logits, pack, hid)
        ctx.mark_non_differentiable(logits)
        ctx.set_materialize_grads(False)
        return loss, logits

    @staticmethod
    def backward(ctx, g_loss, _g_logits):
        z, target, logits, pack, hid = ctx.saved_tensors
        g_loss = _chk(g_loss, "decoder_loss.backward")
        B, _, H, W = z.shape
        C = ctx.C
        w1, b1, w2, b2 = ctx.params
        sinks = [_grad_sink(p) for p in ctx.params]
        acc = all(s is not None for s in sinks)
        grads = sinks if acc else [torch.empty_like(p) for p in ctx.params]
        dz = torch.empty_like(z)
        ws = _new(z, max(1, _decoder_fused_ws(B, H, W, C)[1]))
        d = L.DecoderFusedBwdDesc(_p(z), _p(pack), _p(logits), _p(target), _p(g_loss), _p(dz), _p(grads[0]), _p(grads[1]),
                                  _p(grads[2]), _p(grads[3]), _p(ws), B, H, W, C, ctx.kind, 1 if acc else 0, _p(hid))
        if kernel_timer is not None:
            ev = kernel_timer.bracket("decoder_fused_bwd", 2.0 * B * H * W * 9 * (4 * 16 * 64 + 2 * 64 * C))
            ev[0].record()
        L.check(L.lib().scm_decoder_fused_bwd(ctypes.byref(d), _stream()), "scm_decoder_fused_bwd")
        if kernel_timer is not None:
            ev[1].record()
        if acc:
            return dz, None, None, None, None, None, None, None, None
        return dz, None, grads[0], grads[1], grads[2], grads[3], None, None, None


# ------------------------------------------------------ fused reward head
def reward_fused_pack(w1, b1, w2, b2, latent):
    lib = L.lib()
    ts = [_chk(t.detach(), "reward_fused_pack") for t in (w1, b1, w2, b2)]
    pack = _new(ts[0], lib.scm_reward_fused_pack_floats())
    L.check(lib.scm_reward_fused_pack(_p(ts[0]), _p(ts[1]), _p(ts[2]), _p(ts[3]), latent, ts[2].shape[0], _p(pack),
                                      _stream()), "scm_reward_fused_pack")
    return pack


class RewardConvFn(Function):
    """conv2(leaky_relu(conv1(z))) of the reward head (models.py:235-240) in one pass (scm_reward_fused_fwd / _bwd):
    z (B, 16, H, W) -> o (B, 3R, OH, OW).  The hidden map is recomputed in backward."""

    @staticmethod
    def forward(ctx, z, w1, b1, w2, b2, pack, keep_hidden=True):
        z = _chk(z, "reward_conv")
        B, _, H, W = z.shape
        Q = w2.shape[0]
        o = _new(z, B, Q, (H - 5) // 2 + 1, (W - 5) // 2 + 1)
        if kernel_timer is not None:
            ev = kernel_timer.bracket("reward_fused_fwd", 2.0 * B * 9 * ((H - 2) * (W - 2) * 16 * 32 + o.shape[2] * o.shape[3] * 32 * Q))
            ev[0].record()
        hid = _new(z, B, 32, H, W) if (save_hidden and keep_hidden and any(ctx.needs_input_grad)) else None
        L.check(L.lib().scm_reward_fused_fwd(_p(z), _p(pack), _p(o), _p(hid), B, H, W, Q, _stream()), "scm_reward_fused_fwd")
        if kernel_timer is not None:
            ev[1].record()
        ctx.params = (w1, b1, w2, b2)
        ctx.save_for_backward(z, pack, hid)
        return o

    @staticmethod
    def backward(ctx, g_o):
        z, pack, hid = ctx.saved_tensors
        g_o = _chk(g_o, "reward_conv.backward")
        B, _, H, W = z.shape
        Q = g_o.shape[1]
        sinks = [_grad_sink(p) for p in ctx.params]
        acc = all(s is not None for s in sinks)
        grads = sinks if acc else [torch.empty_like(p) for p in ctx.params]
        dz = torch.empty_like(z)
        n = ctypes.c_size_t()
        L.check(L.lib().scm_reward_fused_workspace(B, H, W, Q, ctypes.byref(n)), "scm_reward_fused_workspace")
        ws = _new(z, max(1, n.value))
        if kernel_timer is not None:
            OH, OW = (H - 5) // 2 + 1, (W - 5) // 2 + 1
            ev = kernel_timer.bracket("reward_fused_bwd", 4.0 * B * 9 * ((H - 2) * (W - 2) * 16 * 32 + OH * OW * 32 * Q))
            ev[0].record()
        L.check(L.lib().scm_reward_fused_bwd(_p(z), _p(pack), _p(g_o), _p(hid), _p(dz), _p(grads[0]), _p(grads[1]), _p(grads[2]),
                                             _p(grads[3]), _p(ws), B, H, W, Q, 1 if acc else 0, _stream()),
                "scm_reward_fused_bwd")
        if kernel_timer is not None:
            ev[1].record()
        if acc:
            return dz, None, None, None, None, None, None
        return dz, grads[0], grads[1], grads[2], grads[3], None, None


_mean_weights = {}


def mean_all(x):
    """Mean over every element of x as a 0-dim tensor (differentiable): a 1 x n linear
    layer with constant weights 1/n on the linear kernel."""
    n = x.numel()
    key = (n, x.device)
    w = _mean_weights.get(key)
    if w is None:
        w = _mean_weights[key] = torch.full((1, n), 1.0 / n, device=x.device, dtype=torch.float32)
    return LinearFn.apply(x.reshape(1, n), w, None, ACT_NONE).reshape(())


class CoordCatFn(Function):
    """cat([x, coord_x, coord_y], 1), coordconv.py:12-14."""

    @staticmethod
    def forward(ctx, x):
        x = _chk(x, "coord_cat")
        B, C, H, W = x.shape
        out = _new(x, B, C + 2, H, W)
        L.check(L.lib().scm_coord_cat(_p(x), _p(out), B, C, H, W, _stream()), "scm_coord_cat")
        ctx.C = C
        return out

    @staticmethod
    def backward(ctx, g):
        return g[:, :ctx.C].contiguous()


def pixel_mse(logit, target):
    return PixelLossFn.apply(logit, target, 1)


def pixel_bce(logit, target):
    return PixelLossFn.apply(logit, target, 0)


class CSRNSweepsFn(Function):
    """The directional sweeps of CSRN.forward (spatial_recurrent.py:55-116) -> the
    concatenated context map (B, 4*ch, H, W) = [above | below | left | right].

    Reference behaviour kept: the right sweep writes context_left (:110), which
    overwrites the left sweep, and context_right stays zero -- so the left sweep
    is not computed and its parameters get zero gradients.
    params: 4 tensors (w_ih, w_hh, conv weight, conv bias) for each of
    down, up, left, right, in that order."""

    SWEEPS = (("down", 1, 0, 0), ("up", 1, 1, 1), ("right", 0, 1, 2))   # name, along_rows, reverse, context slot

    @staticmethod
    def forward(ctx, x, *params):
        lib = L.lib()
        x = _chk(x, "csrn")
        params = [_chk(p.detach(), "csrn") for p in params]
        B, ch, H, W = x.shape
        cmap = torch.zeros(B, 4 * ch, H, W, device=x.device, dtype=torch.float32)
        saved = []
        by_name = {"down": params[0:4], "up": params[4:8], "left": params[8:12], "right": params[12:16]}
        for name, along, rev, slot in CSRNSweepsFn.SWEEPS:
            w_ih, w_hh, wc, bc = by_name[name]
            ntok, nsteps = (W, H) if along else (H, W)
            cbuf = _new(x, B, ch, H, W)
            hprev = _new(x, nsteps, B, ntok, ch)
            d = L.CsrnDesc(_p(x), _p(cbuf), _p(hprev), _p(w_ih), _p(w_hh), _p(wc), _p(bc), B, ch, H, W, along, rev)
            L.check(lib.scm_csrn_sweep_fwd(ctypes.byref(d), _stream()), "scm_csrn_sweep_fwd")
            cmap[:, slot * ch:(slot + 1) * ch] = cbuf
            saved += [cbuf, hprev]
        ctx.geom = (B, ch, H, W)
        ctx.save_for_backward(x, *params, *saved)
        return cmap

    @staticmethod
    def backward(ctx, g):
        lib = L.lib()
        B, ch, H, W = ctx.geom
        t = ctx.saved_tensors
        x, params, saved = t[0], t[1:17], t[17:]
        g = _chk(g, "csrn.backward")
        by_name = {"down": params[0:4], "up": params[4:8], "left": params[8:12], "right": params[12:16]}
        dx = torch.zeros_like(x)
        nparam = 9 * ch * ch + ch
        grads = {"left": [torch.zeros_like(p) for p in by_name["left"]]}
        for i, (name, along, rev, slot) in enumerate(CSRNSweepsFn.SWEEPS):
            w_ih, w_hh, wc, bc = by_name[name]
            cbuf, hprev = saved[2 * i], saved[2 * i + 1]
            dctx = g[:, slot * ch:(slot + 1) * ch].contiguous()
            slab = _new(x, B, nparam)
            dpar = _new(x, nparam)
            d = L.CsrnDesc(_p(x), _p(cbuf), _p(hprev), _p(w_ih), _p(w_hh), _p(wc), _p(bc), B, ch, H, W, along, rev)
            L.check(lib.scm_csrn_sweep_bwd(ctypes.byref(d), _p(dctx), _p(dx), _p(slab), _p(dpar), _stream()),
                    "scm_csrn_sweep_bwd")
            n3 = 3 * ch * ch
            grads[name] = [dpar[0:n3].view(3 * ch, ch), dpar[n3:2 * n3].view(3 * ch, ch),
                           dpar[2 * n3:3 * n3].view(ch, ch, 3), dpar[3 * n3:]]
        out = [dx]
        for name in ("down", "up", "left", "right"):
            out += grads[name]
        return tuple(out)


class StepLossSeqFn(Function):
    """StepLossFn for ALL unrolled steps of an iteration (main.py:177-197 with the decoder / reward head applied to the
    steps' latents in one time-batched launch each, Trainer.step): rhat_all (nT*B, R) and bce_all (nT*B,) hold step
    t0 + i in rows [i*B, (i+1)*B); rewards (B, T, R) / dones (B, T) are the iteration's tensors.  The same per-step kernel
    runs nT times in the reference's order (mask update, both masked means, loss accumulation), reading / writing slices
    in place: no autograd slicing, bitwise the per-step result.  Returns (loss, rd_rec (nT, 2), final mask)."""

    @staticmethod
    def forward(ctx, rhat_all, bce_all, rewards, dones, t0, nT, mask0, coef):
        lib = L.lib()
        rhat_all, bce_all, mask0 = _chk(rhat_all, "step_loss_seq"), _chk(bce_all, "step_loss_seq"), _chk(mask0, "step_loss_seq")
        B, R = rhat_all.shape[0] // nT, rhat_all.shape[1]
        for name, t in (("rewards", rewards), ("dones", dones)):
            if not t.is_cuda or t.dtype != torch.float32:
                raise RuntimeError("scm_gan_amd.step_loss_seq: %s must be an fp32 GPU tensor (got %s on %s)"
                                   % (name, t.dtype, t.device))
        assert rewards.stride(2) == 1 and rewards.shape[0] == B and dones.shape[0] == B and rhat_all.shape[0] == nT * B
        losses, rd_rec = _new(rhat_all, nT), _new(rhat_all, nT, 2)
        masks = _new(rhat_all, nT + 1, B)
        masks[0].copy_(mask0)
        for i in range(nT):
            t = t0 + i
            L.check(lib.scm_step_loss(ctypes.c_void_p(rhat_all.data_ptr() + 4 * i * B * R),
                                      ctypes.c_void_p(rewards.data_ptr() + 4 * t * rewards.stride(1)), rewards.stride(0),
                                      ctypes.c_void_p(dones.data_ptr() + 4 * t * dones.stride(1)), dones.stride(0),
                                      ctypes.c_void_p(bce_all.data_ptr() + 4 * i * B), _p(masks[i]),
                                      None if i == 0 else ctypes.c_void_p(losses.data_ptr() + 4 * (i - 1)), B, R, coef,
                                      _dev("coef"), _p(masks[i + 1]), ctypes.c_void_p(losses.data_ptr() + 4 * i),
                                      _p(rd_rec[i]), _stream()), "scm_step_loss")
        ctx.coef = coef
        ctx.coef_dev = dev_scalars.get("coef")
        ctx.geom = (t0, nT, B, R)
        ctx.save_for_backward(rhat_all, rewards, masks)
        loss, mask_out = losses[nT - 1:nT], masks[nT]
        ctx.mark_non_differentiable(rd_rec, mask_out)
        ctx.set_materialize_grads(False)
        return loss, rd_rec, mask_out

    @staticmethod
    def backward(ctx, g, g_rd_rec, g_mask):
        rhat_all, rewards, masks = ctx.saved_tensors
        t0, nT, B, R = ctx.geom
        g = _chk(g, "step_loss_seq.backward")
        d_rhat, d_bce = torch.empty_like(rhat_all), _new(rhat_all, nT * B)
        for i in range(nT):
            t = t0 + i
            L.check(L.lib().scm_step_loss_bwd(ctypes.c_void_p(rhat_all.data_ptr() + 4 * i * B * R),
                                              ctypes.c_void_p(rewards.data_ptr() + 4 * t * rewards.stride(1)),
                                              rewards.stride(0), _p(masks[i + 1]), _p(g), B, R, ctx.coef, _p(ctx.coef_dev),
                                              ctypes.c_void_p(d_rhat.data_ptr() + 4 * i * B * R),
                                              ctypes.c_void_p(d_bce.data_ptr() + 4 * i * B), _stream()), "scm_step_loss_bwd")
        return d_rhat, d_bce, None, None, None, None, None, None


class StepLossFn(Function):
    """Loss bookkeeping of one unrolled step (main.py:178-197) in one launch; returns
    (loss_out, rd, rec, mask_out).  rewards_t / dones_t may be strided columns."""

    @staticmethod
    def forward(ctx, loss_in, rhat, bce, rewards_t, dones_t, mask_in, coef):
        lib = L.lib()
        rhat, bce, mask_in = _chk(rhat, "step_loss"), _chk(bce, "step_loss"), _chk(mask_in, "step_loss")
        B, R = rhat.shape
        for name, t in (("rewards_t", rewards_t), ("dones_t", dones_t)):
            if not t.is_cuda or t.dtype != torch.float32:
                raise RuntimeError("scm_gan_amd.step_loss: %s must be an fp32 GPU tensor (got %s on %s)"
                                   % (name, t.dtype, t.device))
        assert rewards_t.stride(1) == 1
        loss_out, rd_rec = _new(rhat, 1), _new(rhat, 2)
        mask_out = _new(rhat, B)
        lin = None if loss_in is None else _chk(loss_in, "step_loss")
        L.check(lib.scm_step_loss(_p(rhat), ctypes.c_void_p(rewards_t.data_ptr()), rewards_t.stride(0),
                                  ctypes.c_void_p(dones_t.data_ptr()), dones_t.stride(0), _p(bce), _p(mask_in),
                                  _p(lin), B, R, coef, _dev("coef"), _p(mask_out), _p(loss_out), _p(rd_rec), _stream()),
                "scm_step_loss")
        ctx.coef = coef
        ctx.coef_dev = dev_scalars.get("coef")
        ctx.has_loss_in = loss_in is not None
        ctx.save_for_backward(rhat, rewards_t, mask_out)
        rd, rec = rd_rec[0:1], rd_rec[1:2]
        ctx.mark_non_differentiable(mask_out, rd, rec)
        ctx.set_materialize_grads(False)
        return loss_out, rd, rec, mask_out

    @staticmethod
    def backward(ctx, g, g_rd, g_rec, g_mask):
        rhat, rewards_t, mask = ctx.saved_tensors
        B, R = rhat.shape
        g = _chk(g, "step_loss.backward")
        d_rhat, d_bce = torch.empty_like(rhat), _new(rhat, B)
        L.check(L.lib().scm_step_loss_bwd(_p(rhat), ctypes.c_void_p(rewards_t.data_ptr()), rewards_t.stride(0),
                                          _p(mask), _p(g), B, R, ctx.coef, _p(ctx.coef_dev), _p(d_rhat), _p(d_bce),
                                          _stream()), "scm_step_loss_bwd")
        return (g if ctx.has_loss_in else None), d_rhat, d_bce, None, None, None, None
