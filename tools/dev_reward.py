"""Dev check of the fused reward-head kernels against the layer-by-layer HIP path (GPU box)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import weights as WS
from scm_gan_amd import models as M, ops

def run(B, H, W, R, seed=0):
    torch.manual_seed(seed)
    rp = M.RewardPredictor(16, R)
    rp.load_state_dict(WS.to_torch(WS.reward_state(seed + 3, rewards=R)))
    z0 = (torch.rand(B, 16, H, W, device='cuda') < 0.5).float() * torch.rand(B, 16, H, W, device='cuda')
    gr = torch.randn(B, R, device='cuda')
    out = {}
    for mode in ("1", "0"):
        os.environ["SCM_NO_FUSED_REWARD"] = mode
        z = z0.clone().requires_grad_(True)
        for p in rp.parameters():
            p.grad = None
        r, rmap = rp(z, visualize=True)
        ((r * gr).sum() + 0.01 * rmap.sum()).backward()
        out[mode] = [r.detach().clone(), rmap.detach().clone(), z.grad.clone()] + [p.grad.clone() for p in rp.parameters()]
    torch.cuda.synchronize()
    names = ("r", "map", "dz", "dW1", "db1", "dW2", "db2")
    msg = []
    for nme, a_, r_ in zip(names, out["0"], out["1"]):
        e, m = float((a_ - r_).abs().max()), float(r_.abs().max())
        msg.append("%s %.1e/%.1e" % (nme, e, m))
        bad = int(((a_ - r_).abs() > 1e-4 * max(m, 1e-6) + 1e-9).sum())
        assert bad <= a_.numel() // 50 + 2, (nme, e, m, bad)
    print("B=%d %dx%d R=%d: " % (B, H, W, R) + " ".join(msg))
    return rp, z0

if __name__ == '__main__':
    for shp in [(2, 16, 16, 2), (1, 20, 12, 2), (3, 7, 13, 1), (2, 33, 20, 2), (2, 48, 64, 2), (2, 64, 64, 2), (1, 5, 5, 2), (2, 64, 31, 1)]:
        run(*shp)
    # which path is closer to float64?  (B = 16, CPU autograd in double)
    torch.manual_seed(1)
    rp = M.RewardPredictor(16, 2); rp.load_state_dict(WS.to_torch(WS.reward_state(3)))
    z0 = (torch.rand(16, 16, 64, 64, device='cuda') < 0.5).float()
    import torch.nn.functional as F
    w = [p.detach().double().cpu().requires_grad_(True) for p in rp.parameters()]
    x = F.leaky_relu(F.conv2d(z0.double().cpu(), w[0], w[1]), 0.01)
    x = F.conv2d(x, w[2], w[3], stride=2)
    b, ch, h, ww = x.shape
    x = torch.softmax(x.view(b, 3, ch // 3, h, ww), dim=1)
    (x[:, 0] - x[:, 2]).sum().backward()
    for mode in ("1", "0"):
        os.environ["SCM_NO_FUSED_REWARD"] = mode
        for p in rp.parameters(): p.grad = None
        rp(z0.clone().requires_grad_(True)).sum().backward()
        print("no_fused=%s rel err vs float64:" % mode, ["%.1e" % float((p.grad.double().cpu() - g.grad).abs().max() / g.grad.abs().max()) for p, g in zip(rp.parameters(), w)])
    rp, z0 = run(64, 64, 64, 2)
    for mode in ("1", "0"):
        os.environ["SCM_NO_FUSED_REWARD"] = mode
        z = z0.clone().requires_grad_(True)
        def step():
            rp(z).sum().backward()
        for _ in range(3): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): step()
        torch.cuda.synchronize(); print("no_fused=%s fwd+bwd %.1f us" % (mode, (time.perf_counter() - t0) / 20 * 1e6))
