"""Does the time of a fused-head launch depend on how many launches of that shape precede it?  (tools/bench_heads.py once
showed 4 x 128 samples of the reward head at 3x the time of 1 x 512.)"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from scm_gan_amd import models, ops
dev = torch.device("cuda", 0)
torch.manual_seed(0)
rp = models.RewardPredictor(16, 2)
for p in rp.parameters():
    p.grad = torch.zeros_like(p)
for B in (64, 128, 256):
    z = (torch.rand(B, 16, 64, 64, device=dev) < 0.5).float().requires_grad_(True)
    for reps in (1, 2, 4, 8):
        def f():
            for _ in range(reps):
                with ops.grad_sinks():
                    rp(z).sum().backward()
        for _ in range(3): f()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): f()
        e.record(); torch.cuda.synchronize()
        print("B=%d reps=%d: %.1f us per launch pair" % (B, reps, s.elapsed_time(e) * 100 / reps))
