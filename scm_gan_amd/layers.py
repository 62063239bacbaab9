"""Layer modules backed by the HIP operators (scm_gan_amd.ops).

They keep torch's parameter names, shapes and default initialisation so the
reference's state_dicts load unchanged, but their forward never calls a torch
compute op.
"""
import math

import torch
import torch.nn as nn

from . import ops


def _pair(v):
    return v[0] if isinstance(v, (tuple, list)) else v


class Conv2d(nn.Module):
    """nn.Conv2d(in, out, kernel_size, stride, padding, padding_mode) look-alike.

    3x3 / stride 1 / padding 1 (zero or circular) runs on the MFMA implicit-GEMM
    kernel; any other geometry (RewardPredictor, Discriminator, CoordConv2d
    arguments) on the direct kernel.  `padding=2, padding_mode='circular'` is
    accepted and means one wrapped pixel per side, the torch<=1.4 behaviour the
    reference was written against (models.py:51-56).
    """

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, padding_mode="zeros", bias=True):
        super().__init__()
        k, s, p = _pair(kernel_size), _pair(stride), _pair(padding)
        if padding_mode == "circular":
            if not (k == 3 and s == 1 and p in (1, 2)):
                raise ValueError("circular padding is implemented for the 3x3 stride-1 same-size conv only")
            p = 1
        elif padding_mode != "zeros":
            raise ValueError("unsupported padding_mode %r" % padding_mode)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.padding_mode = k, s, p, padding_mode
        ref = nn.Conv2d(in_channels, out_channels, k, stride=s, padding=p, bias=bias)   # torch default init
        self.weight = nn.Parameter(ref.weight.detach().clone())
        self.bias = nn.Parameter(ref.bias.detach().clone()) if bias else None
        self._pack_key = None
        self._packs = None

    @property
    def is_mfma(self):
        return self.kernel_size == 3 and self.stride == 1 and self.padding == 1

    def _get_packs(self, w, cin_conv):
        key = (w.data_ptr(), w._version, ops.weights_epoch(), cin_conv, w.device, ops.pack_mode())
        if self._pack_key != key:
            self._packs = ops.pack_conv3x3(w, cin_conv, self.out_channels, False, self.in_channels, 0)
            self._pack_key = key
        return self._packs

    def forward(self, x, x2=None, act_vec=None, act=ops.ACT_NONE, _sn=None, dx1_cell=None, dx2_cell=None,
                in_masks=None, pregated=False, mask_holder=None, out_half=False, out_cell=None):
        """x (and optionally x2, concatenated along channels without being
        materialised); act_vec: (B, A) values of A trailing constant input planes
        (circular 3x3 only).  _sn = (weight_bar, u, v, inv_sigma) from SpectralNorm.
        dx1_cell / dx2_cell: shared-gradient cells of x / x2 (ops.SharedGradFork).
        Producer-side gating (ops.pregate_ok): in_masks = the 1-bit leaky-relu gates of x / x2 (this layer's input
        gradient comes out multiplied by them), pregated = every consumer of this layer's output does the same for
        it, mask_holder = a dict that receives this layer's own gate under 'mask'."""
        if _sn is not None:
            w, u, v, inv_sigma = _sn
        else:
            w, u, v, inv_sigma = self.weight, None, None, None
        if self.is_mfma:
            cin_conv = x.shape[1] + (0 if x2 is None else x2.shape[1])
            if act_vec is not None and self.padding_mode != "circular":
                raise ValueError("constant action planes fold into a bias only under circular padding")
            cfg = {"pad": ops.PAD_CIRCULAR if self.padding_mode == "circular" else ops.PAD_ZERO, "act": act,
                   "transposed": False, "packs": self._get_packs(w, cin_conv), "dx1_cell": dx1_cell, "dx2_cell": dx2_cell,
                   "in_masks": in_masks, "pregated": pregated, "mask_holder": mask_holder,
                   # 16-bit operand mode: the output leaves as IEEE half and its fp32 gradient comes back through
                   # out_cell (the dx1_cell / dx2_cell of its consumers): ops.half_store_ok
                   "out_half": out_half, "out_cell": out_cell}
            return ops.Conv3x3Fn.apply(x, x2, act_vec, w, self.bias, inv_sigma, u, v, cfg)
        if x2 is not None or act_vec is not None:
            raise ValueError("two-input / action-plane forms need the 3x3 same-size geometry")
        if self.kernel_size == 3 and self.stride == 1 and self.padding == 0 and x.shape[2] > 2 and x.shape[3] > 2:
            # "valid" 3x3 (RewardPredictor.conv1, models.py:230): the interior of the zero-padded
            # same-size conv, so it runs on the MFMA kernels; the border ring is cropped away
            # (the crop and its zero-embedding backward are plain copies).
            cfg = {"pad": ops.PAD_ZERO, "act": act, "transposed": False, "packs": self._get_packs(w, x.shape[1])}
            full = ops.Conv3x3Fn.apply(x, None, None, w, self.bias, inv_sigma, u, v, cfg)
            return full[:, :, 1:-1, 1:-1].contiguous()
        return ops.Conv2dFn.apply(x, w, self.bias, inv_sigma, u, v, self.kernel_size, self.stride, self.padding, act)

    def forward_same(self, x, act=ops.ACT_NONE):
        """The zero-padded SAME-size 3x3 conv of a padding-0 layer: its interior is this layer's
        'valid' output (the caller reads the interior, e.g. through forward_cropped_input)."""
        assert self.kernel_size == 3 and self.stride == 1 and self.padding == 0
        cfg = {"pad": ops.PAD_ZERO, "act": act, "transposed": False, "packs": self._get_packs(self.weight, x.shape[1])}
        return ops.Conv3x3Fn.apply(x, None, None, self.weight, self.bias, None, None, None, cfg)

    def forward_cropped_input(self, x_full, crop, act=ops.ACT_NONE):
        """This layer applied to x_full[:, :, crop:-crop, crop:-crop] without materialising the
        crop: a padding of -crop on the direct kernel."""
        assert not self.is_mfma
        return ops.Conv2dFn.apply(x_full, self.weight, self.bias, None, None, None, self.kernel_size, self.stride,
                                  self.padding - crop, act)

    def extra_repr(self):
        return "{in_channels}, {out_channels}, kernel_size={kernel_size}, stride={stride}, padding={padding}, " \
               "padding_mode={padding_mode}".format(**self.__dict__)


class ConvTranspose3x3(nn.Module):
    """nn.ConvTranspose2d(in, out, (3,3), stride=1, padding=1) (models.py:260-266):
    weight layout (in, out, 3, 3); computed as the equivalent zero-padded conv with
    channel-transposed, tap-flipped weights on the MFMA kernel."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        ref = nn.ConvTranspose2d(in_channels, out_channels, (3, 3), stride=1, padding=1)
        self.weight = nn.Parameter(ref.weight.detach().clone())
        self.bias = nn.Parameter(ref.bias.detach().clone())
        self._pack_key = None
        self._packs = None

    def forward(self, x, act=ops.ACT_NONE):
        w = self.weight
        key = (w.data_ptr(), w._version, ops.weights_epoch(), w.device, ops.pack_mode())
        if self._pack_key != key:
            self._packs = ops.pack_conv3x3(w, self.in_channels, self.out_channels, True, self.in_channels, 0)
            self._pack_key = key
        cfg = {"pad": ops.PAD_ZERO, "act": act, "transposed": True, "packs": self._packs}
        return ops.Conv3x3Fn.apply(x, None, None, w, self.bias, None, None, None, cfg)


def _convT_factor_sum(self, x, L, C):
    """ConvTranspose3x3 followed by the Decoder's factor sum (models.py:278-287), with
    the sum folded into the weights: out channel c uses sum_f W[:, f*C + c]."""
    w = self.weight                                   # (cin, L*C, 3, 3)
    cin = w.shape[0]
    ws = ops.FactorSumFn.apply(w.reshape(cin, L * C, 3, 3), L, C)                    # (cin, C, 3, 3)
    bs = ops.FactorSumFn.apply(self.bias.reshape(1, L * C, 1, 1), L, C).reshape(C)   # (C,)
    key = (w.data_ptr(), w._version, ops.weights_epoch(), w.device, L, C, ops.pack_mode())
    if getattr(self, "_fs_key", None) != key:           # packs depend on the weight values only
        self._fs_packs = ops.pack_conv3x3(ws, cin, C, True, cin, 0)
        self._fs_key = key
    packs = self._fs_packs
    cfg = {"pad": ops.PAD_ZERO, "act": ops.ACT_NONE, "transposed": True, "packs": packs}
    return ops.Conv3x3Fn.apply(x, None, None, ws, bs, None, None, None, cfg)


ConvTranspose3x3.forward_factor_sum = _convT_factor_sum


class Linear(nn.Module):
    """nn.Linear look-alike for the Discriminator head (models.py:205)."""

    def __init__(self, in_features, out_features):
        super().__init__()
        ref = nn.Linear(in_features, out_features)
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(ref.weight.detach().clone())
        self.bias = nn.Parameter(ref.bias.detach().clone())

    def forward(self, x, act=ops.ACT_NONE):
        return ops.LinearFn.apply(x, self.weight, self.bias, act)
