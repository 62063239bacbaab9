import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from scm_gan_amd import models, ops
dev = torch.device("cuda", 0)
torch.manual_seed(0)
rp = models.RewardPredictor(16, 2)
dec = models.Decoder(16, 3)
for p in list(rp.parameters()) + list(dec.parameters()):
    p.grad = torch.zeros_like(p)
def timed(fn, n=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n
for B in (32, 64, 96, 112, 120, 128, 136, 160, 192, 256, 384, 512, 1024, 2048):
    z = (torch.rand(B, 16, 64, 64, device=dev) < 0.5).float().requires_grad_(True)
    tgt = (torch.rand(B, 3, 64, 64, device=dev) < 0.1).float()
    def f():
        with ops.grad_sinks():
            rp(z).sum().backward()
    def g():
        with ops.grad_sinks():
            dec.forward_loss(z, tgt).sum().backward()
    t, t2 = timed(f), timed(g)
    print("B=%4d reward fwd+bwd %8.1f us = %6.2f us/sample | decoder %8.1f us = %6.2f us/sample" % (B, t, t / B, t2, t2 / B))
