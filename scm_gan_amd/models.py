"""Encoder / Transition / Decoder / RewardPredictor / Discriminator with the
reference's constructor signatures, forward signatures and state_dict keys
(/root/reference/models.py), computed by the HIP operators.

Differences from the reference, all deliberate and documented in DESIGN.md:
  * Transition samples its Bernoulli latent from injected uniform noise or an
    on-device Philox stream instead of torch's global generator
    (SURVEY.md section 8a row B);
  * Discriminator.forward flattens before fc1 (the reference's cannot run,
    SURVEY.md section 0.4);
  * modules move to the GPU in their constructor only when one is visible.
"""
import os

import torch
import torch.nn as nn

from . import ops
from .layers import Conv2d, ConvTranspose3x3, Linear
from .spectral_normalization import SpectralNorm
from .coordconv import CoordConv2d            # noqa: F401  (re-exported like the reference, models.py:15)
from .spatial_recurrent import CSRN           # noqa: F401  (models.py:14)

NOISE_DIM = 3
ENCODER_INPUT_FRAMES = 3


def _to_gpu(module):
    if torch.cuda.is_available():
        module.cuda()


class Transition(nn.Module):
    """models.py:43-119.  conv1..5 spectral-normalised, circular 3x3, U-net skips."""

    def __init__(self, latent_size, num_actions):
        super().__init__()
        self.latent_size = latent_size
        circ = dict(stride=1, padding=2, padding_mode='circular')
        self.conv1 = SpectralNorm(Conv2d(latent_size + num_actions, 128, (3, 3), **circ))
        self.conv2 = SpectralNorm(Conv2d(128, 128, (3, 3), **circ))
        self.conv3 = SpectralNorm(Conv2d(128, 128, (3, 3), **circ))
        self.conv4 = SpectralNorm(Conv2d(128, 128, (3, 3), **circ))
        self.conv5 = SpectralNorm(Conv2d(128 + 128, 128, (3, 3), **circ))
        self.conv6 = Conv2d(128 + 128, latent_size, (3, 3), **circ)
        # Bernoulli sampling state: explicit noise tensors (parity tests) or Philox
        self._noise_queue = []
        self.noise_seed = 0
        self._noise_offset = 0
        _to_gpu(self)

    def set_noise(self, tensors):
        """Queue uniform-noise tensors (B, latent, H, W), consumed one per
        train-mode forward call; when the queue is empty the on-device Philox
        stream (noise_seed, running offset) is used."""
        self._noise_queue = list(tensors)

    def seed_noise(self, seed, offset=0):
        self.noise_seed, self._noise_offset = int(seed), int(offset)

    def forward(self, s, a, return_all=False):
        batch_size, z, height, width = s.shape
        batch_size_actions, num_actions = a.shape
        assert batch_size == batch_size_actions
        L = ops.ACT_LRELU
        # Producer-side gating (ops.pregate_ok): every consumer of the five leaky-relu maps is one of the convolutions
        # below, so each layer's input-gradient kernel gates the gradient it hands down and the layers run their own
        # gradient kernels ungated.  Not when the caller gets the maps themselves (return_all): a gradient arriving
        # from outside would not be gated.
        pg = (torch.is_grad_enabled() and not return_all and ops.pregate_ok(height, width)
              and os.environ.get("SCM_NO_SHARED_GRAD", "0") in ("", "0"))
        m1, m2, m3, m4, m5 = ({} if pg else None for _ in range(5))
        mk = (lambda h: h["mask"]) if pg else (lambda h: None)
        if not return_all and ops.half_store_ok():
            # 16-bit operand mode (training path and rollouts; not when the caller receives the maps): skip2, out3 and
            # out4 -- the maps that only the wide 16-bit kernels read -- cross HBM as IEEE half (ops.half_store_ok); their
            # fp32 gradients come back through one cell per map, into which every consumer's input-gradient convolution
            # writes / accumulates.  skip1 and out5 stay fp32: conv6's thin kernels (16 output channels: forward and the
            # direct weight gradient) read one element per lane and channel and are SLOWER on 2-byte loads (measured).
            g2, g3, g4 = ({} for _ in range(3))
            skip1 = self.conv1(s, None, a, act=L)
            fork = torch.is_grad_enabled() and skip1.requires_grad
            c1 = {} if fork else None
            s1a, s1b = ops.SharedGradFork.apply(skip1, c1) if fork else (skip1, skip1)
            skip2 = self.conv2(s1a, act=L, dx1_cell=c1, out_half=True, out_cell=g2)
            out3 = self.conv3(skip2, act=L, dx1_cell=g2, out_half=True, out_cell=g3)
            out4 = self.conv4(out3, act=L, dx1_cell=g3, out_half=True, out_cell=g4)
            out5 = self.conv5(out4, skip2, act=L, dx1_cell=g4, dx2_cell=g2)
            p = self.conv6(out5, s1b, act=ops.ACT_SIGMOID, dx2_cell=c1)
            return self._sample(p)
        # the action planes (models.py:68-73) enter conv1 as a per-sample bias
        skip1 = self.conv1(s, None, a, act=L, pregated=pg, mask_holder=m1)
        if torch.is_grad_enabled() and skip1.requires_grad and os.environ.get("SCM_NO_SHARED_GRAD", "0") in ("", "0"):
            # the two U-net skips have two consumers each (models.py:78-101): their gradients meet in one buffer
            # inside the gradient convolutions (ops.SharedGradFork) instead of in an elementwise add
            c1, c2 = {}, {}
            s1a, s1b = ops.SharedGradFork.apply(skip1, c1)
            skip2 = self.conv2(s1a, act=L, dx1_cell=c1, in_masks=(mk(m1), None), pregated=pg, mask_holder=m2)
            s2a, s2b = ops.SharedGradFork.apply(skip2, c2)
            out3 = self.conv3(s2a, act=L, dx1_cell=c2, in_masks=(mk(m2), None), pregated=pg, mask_holder=m3)
            out4 = self.conv4(out3, act=L, in_masks=(mk(m3), None), pregated=pg, mask_holder=m4)
            out5 = self.conv5(out4, s2b, act=L, dx2_cell=c2, in_masks=(mk(m4), mk(m2)), pregated=pg,
                              mask_holder=m5)                                          # cat([x, skip2]) never materialised
            p = self.conv6(out5, s1b, act=ops.ACT_SIGMOID, dx2_cell=c1, in_masks=(mk(m5), mk(m1)))   # cat([x, skip1])
        else:
            skip2 = self.conv2(skip1, act=L)
            out3 = self.conv3(skip2, act=L)
            out4 = self.conv4(out3, act=L)
            out5 = self.conv5(out4, skip2, act=L)
            p = self.conv6(out5, skip1, act=ops.ACT_SIGMOID)
        x = self._sample(p)
        if return_all:
            return (skip1, skip2, out3, out4, out5, x)
        return x

    def _sample(self, p):
        """models.py:104-113: Bernoulli straight-through sample in training, threshold in eval."""
        if self.training:
            if self._noise_queue:
                return ops.BernoulliSTFn.apply(p, self._noise_queue.pop(0), 1, 0, 0)
            x = ops.BernoulliSTFn.apply(p, None, 2, self.noise_seed, self._noise_offset)
            self._noise_offset += (p.numel() + 3) // 4
            return x
        return ops.threshold(p)


class Encoder(nn.Module):
    """models.py:123-157."""

    def __init__(self, latent_size, color_channels):
        super().__init__()
        self.latent_size = latent_size
        self.color_channels = color_channels
        self.conv1 = SpectralNorm(Conv2d(color_channels * ENCODER_INPUT_FRAMES, 128, (3, 3), stride=1, padding=1))
        self.bn_conv1 = nn.BatchNorm2d(128)     # registered but never applied (models.py:130,145-155)
        self.conv2 = SpectralNorm(Conv2d(128, 128, (3, 3), stride=1, padding=1))
        self.conv3 = SpectralNorm(Conv2d(128, 128, (3, 3), stride=1, padding=1))
        self.conv4 = Conv2d(128, latent_size, (3, 3), stride=1, padding=1)
        _to_gpu(self)

    def forward(self, x):
        batch_size, frames, channels, height, width = x.shape
        x = x.reshape(batch_size, frames * channels, height, width)
        if ops.half_store_ok():
            # 16-bit operand mode: the two hidden maps between wide layers as IEEE half, gradients through cells (as in
            # Transition); conv3's output feeds the 16-channel conv4 and stays fp32
            g1, g2 = {}, {}
            x = self.conv1(x, act=ops.ACT_LRELU, out_half=True, out_cell=g1)
            x = self.conv2(x, act=ops.ACT_LRELU, dx1_cell=g1, out_half=True, out_cell=g2)
            x = self.conv3(x, act=ops.ACT_LRELU, dx1_cell=g2)
            return self.conv4(x, act=ops.ACT_SIGMOID)
        pg = torch.is_grad_enabled() and ops.pregate_ok(height, width)       # producer-side gating, as in Transition
        m1, m2, m3 = ({} if pg else None for _ in range(3))
        mk = (lambda h: h["mask"]) if pg else (lambda h: None)
        x = self.conv1(x, act=ops.ACT_LRELU, pregated=pg, mask_holder=m1)
        x = self.conv2(x, act=ops.ACT_LRELU, in_masks=(mk(m1), None), pregated=pg, mask_holder=m2)
        x = self.conv3(x, act=ops.ACT_LRELU, in_masks=(mk(m2), None), pregated=pg, mask_holder=m3)
        return self.conv4(x, act=ops.ACT_SIGMOID, in_masks=(mk(m3), None))


class Discriminator(nn.Module):
    """models.py:195-223 (+ the flatten in front of fc1)."""

    def __init__(self):
        super().__init__()
        self.conv1 = SpectralNorm(Conv2d(NOISE_DIM, 32, (3, 3), stride=2, padding=0))
        self.conv2 = SpectralNorm(Conv2d(32, 32, (3, 3), stride=2, padding=0))
        self.conv3 = Conv2d(32, 32, (3, 3), stride=2, padding=0)
        self.fc1 = Linear(32 * 7 * 7, 1)
        _to_gpu(self)

    def forward(self, x):
        x = self.conv1(x, act=ops.ACT_LRELU)
        x = self.conv2(x, act=ops.ACT_LRELU)
        x = self.conv3(x, act=ops.ACT_LRELU)
        x = x.reshape(x.shape[0], -1)
        return self.fc1(x, act=ops.ACT_LRELU)


class RewardPredictor(nn.Module):
    """models.py:226-250."""

    def __init__(self, latent_dim, num_rewards):
        super().__init__()
        self.num_rewards = num_rewards
        self.conv1 = Conv2d(latent_dim, 32, (3, 3), stride=1, padding=0)
        self.conv2 = Conv2d(32, num_rewards * 3, (3, 3), stride=2, padding=0)
        _to_gpu(self)

    def _fused(self, x):
        return x.shape[1] == 16 and self.num_rewards <= 2 and 5 <= x.shape[3] <= 64 and x.shape[2] >= 5 and \
            ops.compute_precision() != "fp16" and os.environ.get("SCM_NO_FUSED_REWARD", "0") in ("", "0")

    def _fused_pack(self):
        ps = (self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias)
        key = (ops.weights_epoch(),) + tuple((p.data_ptr(), p._version) for p in ps)
        if getattr(self, "_fpack_key", None) != key:
            self._fpack = ops.reward_fused_pack(*ps, 16)
            self._fpack_key = key
        return self._fpack

    def forward(self, x, visualize=False):
        if self._fused(x):
            # both convolutions in one pass over z (scm_reward_fused_*), the 32-channel map stays on chip
            x = ops.RewardConvFn.apply(x, self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias,
                                       self._fused_pack(), torch.is_grad_enabled())
        elif x.shape[2] > 2 and x.shape[3] > 2:
            # conv1 (valid 3x3) = interior of the same-size conv on the MFMA kernel; conv2 reads that
            # interior in place (padding -1): no crop copy forward, no zero-embed backward
            x = self.conv1.forward_same(x, act=ops.ACT_LRELU)
            x = self.conv2.forward_cropped_input(x, 1)
        else:
            x = self.conv1(x, act=ops.ACT_LRELU)
            x = self.conv2(x)
        r, rmap = ops.RewardTailFn.apply(x, self.num_rewards)
        if visualize:
            return r, rmap
        return r


class Decoder(nn.Module):
    """models.py:253-291."""

    def __init__(self, latent_size, color_channels):
        super().__init__()
        self.latent_size = latent_size
        self.color_channels = color_channels
        self.conv1 = ConvTranspose3x3(latent_size, latent_size * 4)
        self.conv2 = ConvTranspose3x3(latent_size * 4, latent_size * self.color_channels)
        _to_gpu(self)

    # ---- fused path (scm_decoder_fused_*): one pass over z, the 64-channel hidden map stays on chip.
    # Serves latent_size 16 and <= 3 colour channels in the fp32 mode; anything else takes the layer-by-layer path.
    def _fused(self):
        return self.latent_size == 16 and self.color_channels <= 3 and ops.compute_precision() != "fp16" \
            and os.environ.get("SCM_NO_FUSED_DECODER", "0") in ("", "0")

    def _fused_pack(self):
        ps = (self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias)
        key = (ops.weights_epoch(),) + tuple((p.data_ptr(), p._version) for p in ps)
        if getattr(self, "_fpack_key", None) != key:
            self._fpack = ops.decoder_fused_pack(*ps, self.latent_size, self.color_channels)
            self._fpack_key = key
        return self._fpack

    def forward_loss(self, z_map, target, kind=0):
        """Per-sample pixel loss of the decoded frame against `target` (B, C, H, W) without materialising anything
        but the logits: kind 0 = mean BCE of sigmoid(decoder(z)) (main.py:188-190 + decoder_pixel_loss, :310-312),
        kind 1 = mean squared error (main.py:814-816).  Differentiable w.r.t. z and the decoder's parameters."""
        if not self._fused():
            return ops.PixelLossFn.apply(self.forward(z_map), target, kind)
        loss, _ = ops.DecoderLossFn.apply(z_map, target, self.conv1.weight, self.conv1.bias, self.conv2.weight,
                                          self.conv2.bias, self._fused_pack(), kind, torch.is_grad_enabled())
        return loss

    def forward(self, z_map, visualize=False):
        batch_size, latent_size, height, width = z_map.shape
        if not visualize and not torch.is_grad_enabled() and self._fused():
            # inference (eval rollout, planner frames): the fused forward, logits only
            return ops.decoder_fused_fwd(z_map, self._fused_pack(), self.color_channels)[0]
        x = self.conv1(z_map, act=ops.ACT_LRELU)
        if visualize:
            # the per-factor maps are wanted: materialise conv2's latent_size*C channels (models.py:278-290)
            x = self.conv2(x)
            out = ops.FactorSumFn.apply(x, latent_size, self.color_channels)
            return out, x.view(batch_size, latent_size, self.color_channels, height, width)[0]
        # x.view(B, L, C, H, W).sum(1) of a linear conv == the conv with its weights (and bias) summed
        # over the L factors: conv2 shrinks from 64 -> L*C to 64 -> C output channels, forward and
        # backward (its weight gradient is the broadcast of the summed-weight gradient: FactorSumFn).
        return self.conv2.forward_factor_sum(x, latent_size, self.color_channels)
