"""Inference rollout (transition + decoder, no gradients) at the eval batch of main.py:784-835 (B = 100, 64 x 64): time per
unrolled step -- for A/B runs of kernel switches (SCM_WINO_T2=0 ...)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from scm_gan_amd import models, main as M
torch.manual_seed(0)
B, A = 100, 5
tr, dec = models.Transition(16, A), models.Decoder(16, 3)
M.test_mode([tr, dec])
z = (torch.rand(B, 16, 64, 64, device="cuda") < 0.5).float()
oh = torch.eye(A, device="cuda")[torch.randint(0, A, (B,), device="cuda")]
with torch.no_grad():
    for _ in range(5):
        z = tr(z, oh); dec(z)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50):
        z = tr(z, oh); dec(z)
    e.record(); torch.cuda.synchronize()
print("eval rollout, B=100: %.1f us per step (transition + decoder)" % (s.elapsed_time(e) * 1e3 / 50))
