import sys, os
sys.path.insert(0, 'tools')
import torch
from dev_decoder import run, unfused
from scm_gan_amd import ops
B, H, W, C = [int(v) for v in sys.argv[1:5]]
torch.manual_seed(0)
import weights as WS
from scm_gan_amd import models as M
dec = M.Decoder(16, C)
dec.load_state_dict(WS.to_torch(WS.decoder_state(2, color=C)))
z = ((torch.rand(B, 16, H, W, device='cuda') < 0.5).float() * torch.rand(B, 16, H, W, device='cuda')).requires_grad_(True)
tgt = (torch.rand(B, C, H, W, device='cuda') < 0.2).float()
gl = torch.rand(B, device='cuda') + 0.5
pack = ops.decoder_fused_pack(dec.conv1.weight, dec.conv1.bias, dec.conv2.weight, dec.conv2.bias, 16, C)
ref_loss = ops.PixelLossFn.apply(unfused(dec, z), tgt, 0)
(ref_loss * gl).sum().backward()
rz = z.grad.clone(); z.grad = None
loss, lg = ops.DecoderLossFn.apply(z, tgt, dec.conv1.weight, dec.conv1.bias, dec.conv2.weight, dec.conv2.bias, pack, 0)
(loss * gl).sum().backward()
err = (z.grad - rz).abs()
print("TH env", os.environ.get("SCM_DF_TH"), "max err %.2e of %.2e" % (float(err.max()), float(rz.abs().max())))
rows = err.amax(dim=(0, 1, 3))
print("rows with err > 1e-8:", [int(i) for i in torch.nonzero(rows > 1e-8).flatten()][:40])
cols = err.amax(dim=(0, 1, 2))
print("cols with err > 1e-8:", [int(i) for i in torch.nonzero(cols > 1e-8).flatten()][:40])
bs = err.amax(dim=(1, 2, 3)); print("samples:", [int(i) for i in torch.nonzero(bs > 1e-8).flatten()][:20])
ch = err.amax(dim=(0, 2, 3)); print("channels:", [int(i) for i in torch.nonzero(ch > 1e-8).flatten()][:20])
