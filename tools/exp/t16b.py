import os, sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import weights as WS
from helpers import load
from scm_gan_amd import models as M, ops
fx = load("modules_16.npz")
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
res = {}
for mc in (32, 16):
    ops.WINO_MIN_CIN = mc
    ops.bump_weights_epoch()
    tr = M.Transition(16, 5)
    tr.load_state_dict(WS.to_torch(WS.transition_state(1)))
    tr.train()
    zin = dev(fx["tr_z"]).requires_grad_(True)
    a = dev(fx["tr_a"]).requires_grad_(True)
    tr.set_noise([dev(fx["tr_noise"])])
    outs = tr(zin, a, return_all=True)
    (outs[5] * dev(fx["tr_gz"])).sum().backward()
    res[mc] = [o.detach().clone() for o in outs[:5]] + [zin.grad.clone(), a.grad.clone()] + [p.grad.clone() for k, p in tr.named_parameters() if p.grad is not None]
names = ["skip1", "skip2", "out3", "out4", "out5", "gz", "ga"] + [k for k, p in tr.named_parameters() if p.grad is not None]
for n, x, y in zip(names, res[32], res[16]):
    d = (x - y).abs()
    print("%-28s maxdiff %.3e  scale %.3e  nbad %d of %d" % (n, float(d.max()), float(x.abs().max()), int((d > 1e-5 * max(1e-9, float(x.abs().max()))).sum()), d.numel()))
gz = (res[32][5] - res[16][5]).abs()
idx = (gz > 1e-4).nonzero()
print(idx[:20].tolist(), gz.shape)
