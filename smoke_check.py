"""smoke(): one small encoder -> fused decoder + loss -> transition step with backward
on cuda:0 through the C ABI, checked against the CPU oracle.  Lives next to __graft_entry__.py, outside
the product package: nothing under scm_gan_amd/ imports the oracle."""
import os
import sys

import numpy as np
import torch


def smoke():
    root = os.path.dirname(os.path.abspath(__file__))
    for p in (root, os.path.join(root, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import weights as WS
    from oracle import scm_oracle as O
    from scm_gan_amd import models, ops
    assert torch.cuda.is_available(), "smoke() needs cuda:0"
    torch.cuda.set_device(0)
    B, H, W, A = 2, 16, 16, 5
    rng = np.random.default_rng(7)
    frames = (rng.random((B, 3, 3, H, W)) < 0.2).astype(np.float32)
    target = (rng.random((B, 3, H, W)) < 0.2).astype(np.float32)
    act = np.zeros((B, A), np.float32)
    act[np.arange(B), rng.integers(0, A, B)] = 1
    noise = rng.random((B, 16, H, W), dtype=np.float32)
    # ---- HIP path
    enc, tr, dec = models.Encoder(16, 3), models.Transition(16, A), models.Decoder(16, 3)
    enc.load_state_dict(WS.to_torch(WS.encoder_state(0)))
    tr.load_state_dict(WS.to_torch(WS.transition_state(1)))
    dec.load_state_dict(WS.to_torch(WS.decoder_state(2)))
    tr.set_noise([torch.from_numpy(noise).cuda()])
    z = enc(torch.from_numpy(frames).cuda())
    loss = dec.forward_loss(z, torch.from_numpy(target).cuda()).mean()      # fused decoder + BCE (scm_decoder_fused_*)
    z2 = tr(z, torch.from_numpy(act).cuda())
    (loss + (z2 * 0.01).sum()).backward()
    # ---- oracle
    oenc, otr, odec = (O.clone_state(s) for s in (WS.encoder_state(0), WS.transition_state(1), WS.decoder_state(2)))
    for st in (oenc, otr, odec):
        for k in O.trainable(st):
            st[k].requires_grad_(True)
    oz = O.encoder_forward(oenc, torch.from_numpy(frames))
    oloss = O.decoder_pixel_loss(torch.from_numpy(target), torch.sigmoid(O.decoder_forward(odec, oz))).mean()
    oz2 = O.transition_forward(otr, oz, torch.from_numpy(act), True, torch.from_numpy(noise))
    (oloss + (oz2 * 0.01).sum()).backward()
    err_z = float((z.detach().cpu() - oz.detach()).abs().max())
    err_l = abs(float(loss.detach()) - float(oloss.detach()))
    flips = int((z2.detach().cpu() != oz2.detach()).sum())
    g = enc.conv2.module.weight_bar.grad.cpu()
    og = oenc["conv2.module.weight_bar"].grad
    err_g = float((g - og).abs().max() / (og.abs().max() + 1e-12))
    print("smoke: max|z-z_ref|=%.2e |loss-loss_ref|=%.2e latent flips=%d rel grad err=%.2e" % (err_z, err_l, flips, err_g))
    assert err_z < 1e-4 and err_l < 1e-4 and flips <= 2 and err_g < 1e-3, "HIP path disagrees with the oracle"
