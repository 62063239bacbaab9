"""GPU (-m gpu): the rows SURVEY.md section 8(f) marks "next" -- optional loss terms
of train() against the reference's own run with those flags, the MPC rollout
(compute_rollout_reward) and multi-step input gradients (excitation-BPTT style)
against the oracle."""
import numpy as np
import pytest
import torch

import weights as WS
from helpers import load

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def build(seed, C=3, A=5, R=2, L=16):
    from scm_gan_amd import models as M
    enc, dec, rp, disc, tr = M.Encoder(L, C), M.Decoder(L, C), M.RewardPredictor(L, R), M.Discriminator(), \
        M.Transition(L, A)
    enc.load_state_dict(WS.to_torch(WS.encoder_state(seed)))
    tr.load_state_dict(WS.to_torch(WS.transition_state(seed + 1)))
    dec.load_state_dict(WS.to_torch(WS.decoder_state(seed + 2)))
    rp.load_state_dict(WS.to_torch(WS.reward_state(seed + 3)))
    return enc, dec, rp, disc, tr


@pytest.mark.parametrize("case,flags", [
    ("lo", ["--latent-overshooting"]),
    ("cf", ["--enable-disentanglement-loss", "--enable-action-control-loss", "--counterfactual-horizon", "3"]),
])
def test_optional_losses_vs_reference_run(case, flags):
    """main.py:217-234 (latent overshooting) and :242-283 (counterfactual regularisers):
    losses and final weights of the reference's own train() with the same flags."""
    from scm_gan_amd import main as T
    from scm_gan_amd.datasource import SyntheticDatasource
    fx = load("train_flags.npz")
    B, H, W, horizon, iters, _ = [int(v) for v in fx[case + ".cfg"]]
    seed, A, R, C, L = 0, 5, 2, 3, 16
    enc, dec, rp, disc, tr = build(seed + 30)
    nrng = np.random.default_rng(7000 + seed)
    tr.set_noise([dev(nrng.random((int(b), L, H, W), dtype=np.float32)) for b in fx[case + ".noise_shapes"]])
    ds = SyntheticDatasource(seed=seed + 5, height=H, width=W, channels=C, num_actions=A, num_rewards=R, done_every=3)
    T.args = T.parser.parse_args(["--batch-size", str(B), "--train-iters", str(iters), "--horizon-min", str(horizon),
                                  "--horizon-max", str(horizon), "--eval-every", "1000000"] + flags)
    np.random.seed(1234)                  # the CF losses draw from numpy's global stream (main.py:249-250,275)
    ts = T.train(L, ds, A, R, enc, dec, rp, disc, tr)
    T.args = T.parser.parse_args([])
    names = list(fx[case + ".loss_names"])
    cursor = {}
    got = []
    for n in names:
        i = cursor.get(n, 0)
        got.append(ts.values(n)[i])
        cursor[n] = i + 1
    got = np.array(got)
    ref = fx[case + ".loss_values"]
    assert np.allclose(got, ref, atol=3e-4, rtol=3e-4), (np.abs(got - ref).max(), names[int(np.abs(got - ref).argmax())])
    for name, m in (("enc", enc), ("tr", tr), ("dec", dec), ("rp", rp)):
        for k, v in m.state_dict().items():
            if not v.dtype.is_floating_point:
                continue
            e = fx["%s.final.%s.%s" % (case, name, k)]
            d = WS.digest(v.detach().cpu().numpy())
            err = np.abs(d[3:] - e[3:])
            assert np.mean(err < 1e-5) > 0.95 and err.max() < 1.2e-3, (case, name, k, err.max(), np.mean(err < 1e-5))


def test_mpc_rollout_vs_oracle():
    """compute_rollout_reward (main.py:455-489): beam of A^2 action sequences x depth 12."""
    from scm_gan_amd import main as T
    from oracle import scm_oracle as O
    enc, dec, rp, disc, tr = build(40)
    T.test_mode([enc, dec, rp, disc, tr])
    rng = np.random.default_rng(3)
    z0 = (rng.random((1, 16, 16, 16)) < 0.5).astype(np.float32)
    with torch.no_grad():
        best = T.compute_rollout_reward(dev(z0), tr, rp, 5, selected_action=0)
    otr, orp = O.clone_state(WS.transition_state(41)), O.clone_state(WS.reward_state(43))
    obest, _ = O.compute_rollout_reward(torch.from_numpy(z0), otr, orp, 5)
    assert abs(float(best) - float(obest)) < 5e-3 * max(1.0, abs(float(obest))), (float(best), float(obest))


def test_multistep_input_gradients_vs_oracle():
    """excitation_bptt.py:6-53 pattern: unroll transitions keeping gradients on z_0 and
    every action vector, backprop from the reward map."""
    from oracle import scm_oracle as O
    enc, dec, rp, disc, tr = build(50)
    for m in (tr, rp):
        m.train()
    rng = np.random.default_rng(5)
    B, H, W, A, steps = 2, 16, 16, 5, 3
    z0 = rng.random((B, 16, H, W)).astype(np.float32)
    acts = [rng.random((B, A)).astype(np.float32) for _ in range(steps)]
    noise = [rng.random((B, 16, H, W), dtype=np.float32) for _ in range(steps)]
    gmap = rng.standard_normal((B, 2, (H - 2 - 3) // 2 + 1, (W - 2 - 3) // 2 + 1)).astype(np.float32)
    # HIP
    z = dev(z0).requires_grad_(True)
    a_t = [dev(a).requires_grad_(True) for a in acts]
    tr.set_noise([dev(n) for n in noise])
    zz = z
    for t in range(steps):
        zz = tr(zz, a_t[t])
    r, rmap = rp(zz, visualize=True)
    (rmap * dev(gmap)).sum().backward()
    # oracle
    otr, orp = O.clone_state(WS.transition_state(51)), O.clone_state(WS.reward_state(53))
    oz = torch.from_numpy(z0).requires_grad_(True)
    oa = [torch.from_numpy(a).requires_grad_(True) for a in acts]
    ozz = oz
    for t in range(steps):
        ozz = O.transition_forward(otr, ozz, oa[t], True, torch.from_numpy(noise[t]))
    _, ormap = O.reward_forward(orp, ozz, visualize=True)
    (ormap * torch.from_numpy(gmap)).sum().backward()
    assert int((zz.detach().cpu() != ozz.detach()).sum()) <= 2
    gz, ogz = z.grad.cpu(), oz.grad
    assert float((gz - ogz).abs().max()) < 2e-4 * max(1.0, float(ogz.abs().max()))
    for t in range(steps):
        ga, oga = a_t[t].grad.cpu(), oa[t].grad
        assert float((ga - oga).abs().max()) < 1e-3 * max(1.0, float(oga.abs().max())), t
