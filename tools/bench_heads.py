#!/usr/bin/env python3
"""Fused decoder + reward head, forward and backward: 8 launches of 64 samples (one per unrolled step, as the
reference's loop issues them) against ONE launch of 512 samples (all steps of an iteration at once).

  python tools/bench_heads.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scm_gan_amd import models, ops      # noqa: E402


def timed(fn, n=10):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    H = W = int(os.environ.get("SIZE", "64"))
    dec, rp = models.Decoder(16, 3), models.RewardPredictor(16, 2)
    for p in list(dec.parameters()) + list(rp.parameters()):
        p.grad = torch.zeros_like(p)
    for B, reps in ((64, 8), (128, 4), (256, 2), (512, 1)):
        z = (torch.rand(B, 16, H, W, device=dev) < 0.5).float().requires_grad_(True)
        tgt = (torch.rand(B, 3, H, W, device=dev) < 0.1).float()

        def dec_f():
            with torch.no_grad():
                for _ in range(reps):
                    dec.forward_loss(z, tgt)

        def rp_f():
            with torch.no_grad():
                for _ in range(reps):
                    rp(z)

        def dec_fb():
            for _ in range(reps):
                with ops.grad_sinks():
                    dec.forward_loss(z, tgt).sum().backward()

        def rp_fb():
            for _ in range(reps):
                with ops.grad_sinks():
                    rp(z).sum().backward()

        print("B=%3d x %d launches: decoder fwd %7.1f us, fwd+bwd %7.1f us | reward head fwd %7.1f us, fwd+bwd %7.1f us"
              % (B, reps, timed(dec_f), timed(dec_fb), timed(rp_f), timed(rp_fb)))


if __name__ == "__main__":
    main()
