import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import weights as WS
from scm_gan_amd import models as M, ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dec = M.Decoder(16, 3)
z = (torch.rand(B, 16, 64, 64, device='cuda') < 0.5).float()
tgt = (torch.rand(B, 3, 64, 64, device='cuda') < 0.2).float()
pack = ops.decoder_fused_pack(dec.conv1.weight, dec.conv1.bias, dec.conv2.weight, dec.conv2.bias, 16, 3)
for _ in range(30):
    ops.decoder_fused_fwd(z, pack, 3, tgt, 0, True)
torch.cuda.synchronize()
