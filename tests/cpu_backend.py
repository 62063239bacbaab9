"""TEST INFRASTRUCTURE -- never imported by the product.

Lets the HOST logic of scm_gan_amd.main (Trainer.step / train(): flat gradient buckets, the
autograd hook that issues the early all-reduces, stream ordering, optimizer order, evaluation
cadence, rank seeding) run on CPU ranks under gloo, where no HIP kernel can launch:

  * the networks are torch-CPU modules whose forward is the ORACLE's arithmetic
    (oracle/scm_oracle.py), with the reference's state_dict keys;
  * the handful of HIP leaf operators main.py calls directly (step loss, pixel losses, fused
    clip+Adam, mean, U{0,1} sampling) are monkeypatched with torch-CPU equivalents;
  * torch.cuda's stream / event API is replaced by no-op stand-ins.

The product itself has no CPU path: scm_gan_amd.ops raises on CPU tensors.
"""
import contextlib
import math

import numpy as np
import torch
import torch.nn as nn

from oracle import scm_oracle as O


# ------------------------------------------------------------------ torch.cuda stand-ins
class DummyStream(object):
    cuda_stream = 0

    def __init__(self, *a, **k):
        pass

    def wait_stream(self, other):
        pass

    def wait_event(self, ev):
        pass

    def synchronize(self):
        pass


class DummyEvent(object):
    def __init__(self, *a, **k):
        pass

    def record(self, stream=None):
        pass

    def synchronize(self):
        pass


# ------------------------------------------------------------------------- leaf operators
class StepLoss(object):
    """ops.StepLossFn.apply: main.py:178-197."""

    @staticmethod
    def apply(loss_in, rhat, bce, rewards_t, dones_t, mask_in, coef):
        mask = mask_in * (1 - dones_t)
        rd = torch.mean(torch.mean((rhat - rewards_t) ** 2, dim=1) * mask)
        rec = torch.mean(bce * mask)
        loss = coef * rd + rec
        if loss_in is not None:
            loss = loss + loss_in
        return loss, rd.detach(), rec.detach(), mask.detach()


class StepLossSeq(object):
    """ops.StepLossSeqFn.apply: the per-step bookkeeping over the time-batched heads' outputs."""

    @staticmethod
    def apply(rhat_all, bce_all, rewards, dones, t0, nT, mask0, coef):
        B = rhat_all.shape[0] // nT
        loss, mask, rows = None, mask0, []
        for i in range(nT):
            loss, rd, rec, mask = StepLoss.apply(loss, rhat_all[i * B:(i + 1) * B], bce_all[i * B:(i + 1) * B],
                                                 rewards[:, t0 + i], dones[:, t0 + i], mask, coef)
            rows.append(torch.stack([rd, rec]))
        return loss, torch.stack(rows), mask


def pixel_bce(logit, target):
    return O.decoder_pixel_loss(target, torch.sigmoid(logit))


def pixel_mse(logit, target):
    return ((target - torch.sigmoid(logit)) ** 2).mean(-1).mean(-1).mean(-1)


class AdamClipCPU(object):
    """optim.AdamClip: grad * grad_scale -> clamp(+-clip) -> torch.optim.Adam defaults."""

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, clip_value=0.0):
        self.params = list(params)
        self.lr, self.betas, self.eps, self.clip_value = lr, betas, eps, clip_value
        self.grad_scale = 1.0
        self.state = {}

    def step(self):
        b1, b2 = self.betas
        with torch.no_grad():
            for p in self.params:
                if p.grad is None:
                    continue
                st = self.state.setdefault(id(p), {"t": 0, "m": torch.zeros_like(p), "v": torch.zeros_like(p)})
                g = p.grad * self.grad_scale
                if self.clip_value > 0:
                    g = g.clamp(-self.clip_value, self.clip_value)
                st["t"] += 1
                t = st["t"]
                st["m"].mul_(b1).add_(g, alpha=1 - b1)
                st["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
                denom = (st["v"].sqrt() / math.sqrt(1 - b2 ** t)).add_(self.eps)
                p.addcdiv_(st["m"], denom, value=-self.lr / (1 - b1 ** t))


# ------------------------------------------------------------------------------- networks
class _OracleNet(nn.Module):
    """Parameters under the reference's state_dict keys ('.' -> '__' inside the ParameterDict)."""

    def __init__(self, np_state):
        super().__init__()
        self.p = nn.ParameterDict()
        self.keys = []
        for k, v in np_state.items():
            t = torch.from_numpy(np.array(v)).clone()
            if not t.dtype.is_floating_point:
                continue
            trainable = not (k.endswith("weight_u") or k.endswith("weight_v") or "running_" in k or k.startswith("bn_"))
            self.p[k.replace(".", "__")] = nn.Parameter(t, requires_grad=trainable)
            self.keys.append(k)

    def state(self):
        return {k: self.p[k.replace(".", "__")] for k in self.keys}


class Encoder(_OracleNet):
    def forward(self, x):
        return O.encoder_forward(self.state(), x)


class Transition(_OracleNet):
    noise_seed = 0

    def __init__(self, np_state):
        super().__init__(np_state)
        self._noise = []

    def set_noise(self, tensors):
        self._noise = list(tensors)

    def seed_noise(self, seed, offset=0):
        self.noise_seed = int(seed)

    def forward(self, z, a):
        if self.training:
            return O.transition_forward(self.state(), z, a, True, self._noise.pop(0))
        return O.transition_forward(self.state(), z, a, False)


class Decoder(_OracleNet):
    def forward(self, z):
        return O.decoder_forward(self.state(), z)


class RewardPredictor(_OracleNet):
    def forward(self, z):
        return O.reward_forward(self.state(), z)


class Discriminator(_OracleNet):
    def forward(self, x):
        return O.discriminator_forward(self.state(), x)


def build_nets(seed, A=5, R=2, C=3, L=16):
    import weights as WS
    return (Encoder(WS.encoder_state(seed, L, C)), Decoder(WS.decoder_state(seed + 2, L, C)),
            RewardPredictor(WS.reward_state(seed + 3, L, R)), Discriminator(WS.discriminator_state(seed + 4)),
            Transition(WS.transition_state(seed + 1, L, A)))


# -------------------------------------------------------------------------------- install
def install(mp=None):
    """Patch scm_gan_amd.main's HIP leaves and torch.cuda's stream API.  `mp` = pytest's
    monkeypatch, or None inside a spawned worker process (patched for the process lifetime)."""
    import scm_gan_amd.main as M
    import scm_gan_amd.ops as ops

    def setattr_(obj, name, value):
        if mp is not None:
            mp.setattr(obj, name, value, raising=False)
        else:
            setattr(obj, name, value)

    setattr_(torch.cuda, "Stream", DummyStream)
    setattr_(torch.cuda, "Event", DummyEvent)
    setattr_(torch.cuda, "current_stream", lambda *a, **k: DummyStream())
    setattr_(torch.cuda, "stream", lambda s: contextlib.nullcontext())
    setattr_(M, "_device", lambda: torch.device("cpu"))
    setattr_(M, "AdamClip", AdamClipCPU)
    setattr_(M, "decoder_pixel_loss", lambda target, logits: pixel_bce(logits, target))
    setattr_(ops, "StepLossFn", StepLoss)
    setattr_(ops, "StepLossSeqFn", StepLossSeq)
    setattr_(ops, "pixel_mse", pixel_mse)
    setattr_(ops, "pixel_bce", pixel_bce)
    setattr_(ops, "mean_all", lambda x: x.mean())
    setattr_(ops, "bernoulli_sample", lambda p, seed, off: (torch.rand(p.shape, generator=torch.Generator().manual_seed(
        int(seed) * 7919 + int(off))) < p).float())
    return M
