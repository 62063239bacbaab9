"""CPU: pin the oracle (oracle/scm_oracle.py) against the golden vectors made by
the reference's own modules and train()/measure_prediction_mse()
(tests/golden/make_golden.py)."""
import numpy as np
import pytest
import torch

import weights as WS
from helpers import load, check
from oracle import scm_oracle as O

TOL = 2e-6   # fp32 CPU restatement vs fp32 CPU reference: summation-order noise only


@pytest.mark.parametrize("tag,seed,grads", [("16", 0, True), ("20x12", 10, True), ("64", 20, False)])
def test_modules(tag, seed, grads):
    fx = load("modules_%s.npz" % tag)
    # ---- encoder: models.py:139-157
    enc = O.clone_state(WS.encoder_state(seed))
    for k in O.trainable(enc):
        enc[k].requires_grad_(True)
    x = torch.from_numpy(fx["enc_x"])
    z = O.encoder_forward(enc, x)
    check(fx, "enc_z", z, TOL)
    if grads:
        (z * torch.from_numpy(fx["enc_gz"])).sum().backward()
        for k in O.trainable(enc):
            if k.startswith("bn_"):
                assert enc[k].grad is None
                continue
            check(fx, "enc_grad." + k, enc[k].grad, 2e-5, 1e-4, "encoder")
    for k in enc:
        if k.endswith("weight_u") or k.endswith("weight_v"):
            check(fx, "enc_after1." + k, enc[k], TOL)
    check(fx, "enc_z_call2", O.encoder_forward(enc, x), TOL)

    # ---- transition: models.py:59-119
    tr = O.clone_state(WS.transition_state(seed + 1))
    for k in O.trainable(tr):
        tr[k].requires_grad_(True)
    zin = torch.from_numpy(fx["tr_z"]).requires_grad_(True)
    a = torch.from_numpy(fx["tr_a"]).requires_grad_(True)
    outs = O.transition_forward(tr, zin, a, True, torch.from_numpy(fx["tr_noise"]), return_all=True,
                                return_logits=True)
    for name, o in zip(("skip1", "skip2", "out3", "out4", "out5", "znext", "logits"), outs):
        check(fx, "tr_" + name, o, 5e-6 if name == "logits" else TOL)
    if grads:
        (outs[5] * torch.from_numpy(fx["tr_gz"])).sum().backward()
        check(fx, "tr_grad_z", zin.grad, 2e-5, 1e-4)
        check(fx, "tr_grad_a", a.grad, 2e-4, 1e-4)
        for k in O.trainable(tr):
            check(fx, "tr_grad." + k, tr[k].grad, 5e-5, 1e-4, "transition")
    for k in tr:
        if k.endswith("weight_u") or k.endswith("weight_v"):
            check(fx, "tr_after1." + k, tr[k], TOL)
    with torch.no_grad():
        zn, lg = O.transition_forward(tr, torch.from_numpy(fx["tr_eval_z"]), torch.from_numpy(fx["tr_a"]), False,
                                      return_logits=True)
    check(fx, "tr_eval_logits", lg, 5e-6)
    assert np.array_equal(zn.numpy(), fx["tr_eval_znext"])

    # ---- decoder: models.py:270-291
    dec = O.clone_state(WS.decoder_state(seed + 2))
    for k in dec:
        dec[k].requires_grad_(True)
    zd = torch.from_numpy(fx["dec_z"]).requires_grad_(True)
    y, vis = O.decoder_forward(dec, zd, visualize=True)
    check(fx, "dec_y", y, 5e-6)
    check(fx, "dec_vis", vis, 5e-6)
    if grads:
        (y * torch.from_numpy(fx["dec_gy"])).sum().backward()
        check(fx, "dec_grad_z", zd.grad, 2e-5, 1e-4)
        for k in dec:
            check(fx, "dec_grad." + k, dec[k].grad, 1e-4, 1e-4, "decoder")

    # ---- reward predictor: models.py:235-250
    rp = O.clone_state(WS.reward_state(seed + 3))
    for k in rp:
        rp[k].requires_grad_(True)
    zr = torch.from_numpy(fx["rp_z"]).requires_grad_(True)
    r, rmap = O.reward_forward(rp, zr, visualize=True)
    check(fx, "rp_r", r, 2e-5, 1e-5)
    check(fx, "rp_map", rmap, TOL)
    if grads:
        (r * torch.from_numpy(fx["rp_gr"])).sum().backward()
        check(fx, "rp_grad_z", zr.grad, 2e-6, 1e-4)
        for k in rp:
            check(fx, "rp_grad." + k, rp[k].grad, 2e-5, 1e-4, "reward")


def test_operators():
    fx = load("operators.npz")
    # ---- SpectralNorm call sequence: spectral_normalization.py:23-35,66-68
    st = O.clone_state({k.replace("conv2.", "c."): v for k, v in WS.transition_state(0).items()
                        if k.startswith("conv2.")})
    st["c.module.weight_bar"].requires_grad_(True)
    st["c.module.bias"].requires_grad_(True)
    x = torch.from_numpy(fx["sn_x"])
    for call in (1, 2, 3):
        w, sigma = O.sn_weight(st, "c")
        y = O.conv3x3(x, w, st["c.module.bias"], False)
        check(fx, "sn_u%d" % call, st["c.module.weight_u"], TOL)
        check(fx, "sn_v%d" % call, st["c.module.weight_v"], TOL)
        assert abs(float(sigma.detach()) - float(fx["sn_sigma%d" % call])) < 1e-5
        check(fx, "sn_y%d" % call, y, 1e-5)
    (y * torch.from_numpy(fx["sn_gy"])).sum().backward()
    check(fx, "sn_grad_wbar", st["c.module.weight_bar"].grad, 2e-5, 1e-4)
    check(fx, "sn_grad_bias", st["c.module.bias"].grad, 2e-5, 1e-5)

    # ---- Discriminator with the documented flatten: models.py:208-223
    disc = O.clone_state(WS.discriminator_state(4))
    for k in O.trainable(disc):
        disc[k].requires_grad_(True)
    xd = torch.from_numpy(fx["disc_x"]).requires_grad_(True)
    logit = O.discriminator_forward(disc, xd)
    check(fx, "disc_logit", logit, 1e-5)
    (logit * torch.from_numpy(fx["disc_gl"])).sum().backward()
    check(fx, "disc_grad_x", xd.grad, 1e-6, 1e-4)
    for k in O.trainable(disc):
        check(fx, "disc_grad." + k, disc[k].grad, 2e-5, 1e-4, "discriminator")

    # ---- CoordConv2d: coordconv.py:10-15
    cc = O.clone_state(WS.coordconv_state(5, 6, 10))
    for k in cc:
        cc[k].requires_grad_(True)
    xc = torch.from_numpy(fx["cc_x"]).requires_grad_(True)
    yc = O.coordconv_forward(cc, xc, padding=1)
    check(fx, "cc_y", yc, 5e-6)
    (yc * torch.from_numpy(fx["cc_gy"])).sum().backward()
    check(fx, "cc_grad_x", xc.grad, 1e-5)
    check(fx, "cc_grad.conv.weight", cc["conv.weight"].grad, 5e-5, 1e-5)

    # ---- CSRN incl. the :110 overwrite: spatial_recurrent.py:46-119
    for tag, scale in (("soft", 0.5), ("sat", 8.0)):
        cs = O.clone_state(WS.csrn_state(6, 8, scale))
        for k in cs:
            cs[k].requires_grad_(True)
        xs = torch.from_numpy(fx["csrn_%s_x" % tag]).requires_grad_(True)
        ys = O.csrn_forward(cs, xs)
        # "sat" uses N(0, 8) recurrent weights like the reference init N(0, channels)
        # (spatial_recurrent.py:11-18): the saturated recurrence amplifies fp32
        # rounding differences between GRU formulations ~1000x, hence the looser bound.
        ytol, gtol = (5e-6, 5e-5) if tag == "soft" else (1e-3, 2e-3)
        check(fx, "csrn_%s_y" % tag, ys, ytol)
        (ys * torch.from_numpy(fx["csrn_%s_gy" % tag])).sum().backward()
        def scaled(key):   # tolerance relative to the tensor's largest magnitude
            return gtol * max(1.0, float(np.abs(fx[key]).max()))
        check(fx, "csrn_%s_grad_x" % tag, xs.grad, scaled("csrn_%s_grad_x" % tag))
        for k in cs:
            g = cs[k].grad if cs[k].grad is not None else torch.zeros_like(cs[k])
            key = "csrn_%s_grad.%s" % (tag, k)
            check(fx, key, g, scaled(key), 0.0, "csrn")


def test_train_and_rollout():
    """The oracle's restated train step / eval rollout against the reference's
    own main.train() / measure_prediction_mse() run (3 iterations, B=4, 16x16)."""
    from scm_gan_amd.datasource import SyntheticDatasource
    fx = load("train_rollout.npz")
    B, H, W, horizon, iters, A, R, C, L, seed = [int(v) for v in fx["cfg"]]
    enc = O.clone_state(WS.encoder_state(seed))
    tr = O.clone_state(WS.transition_state(seed + 1))
    dec = O.clone_state(WS.decoder_state(seed + 2))
    rp = O.clone_state(WS.reward_state(seed + 3))
    trainer = O.Trainer(enc, tr, dec, rp, A)
    ds = SyntheticDatasource(seed=seed, height=H, width=W, channels=C, num_actions=A, num_rewards=R, done_every=4)
    nrng = np.random.default_rng(5000 + seed)
    logs = []
    for it in range(1, iters + 1):
        states, rewards, dones, actions = ds.get_trajectories(B, horizon)
        noises = [torch.from_numpy(nrng.random((B, L, H, W), dtype=np.float32)) for _ in range(horizon - 2)]
        _, lg = trainer.step(states, rewards, dones, actions, it / iters, noises)
        logs += lg
        if it == 1:
            for name, st in (("enc", enc), ("tr", tr), ("dec", dec), ("rp", rp)):
                for k in O.trainable(st):
                    if st[k].grad is None:
                        continue
                    e = fx["grad1.%s.%s" % (name, k)]
                    d = WS.digest(st[k].grad.numpy())
                    assert np.allclose(d[3:], e[3:], atol=2e-6, rtol=2e-3), (name, k, np.abs(d[3:] - e[3:]).max())
    names = [n for n, _ in logs]
    assert names == list(fx["train_loss_names"])
    got = np.array([v for _, v in logs])
    assert np.allclose(got, fx["train_loss_values"], atol=1e-5, rtol=1e-5), np.abs(got - fx["train_loss_values"]).max()
    for name, st in (("enc", enc), ("tr", tr), ("dec", dec), ("rp", rp)):
        for k, v in st.items():
            if not v.dtype.is_floating_point:
                continue
            e = fx["final.%s.%s" % (name, k)]
            d = WS.digest(v.detach().numpy())
            # Adam moves every weight by ~lr per step; a sign flip of a ~0 gradient shows as 2e-4
            assert np.allclose(d[3:], e[3:], atol=3e-6, rtol=1e-5), (name, k, np.abs(d[3:] - e[3:]).max())
    # ---- eval rollout, main.py:784-835
    ds_eval = SyntheticDatasource(seed=seed + 77, height=H, width=W, channels=C, num_actions=A, num_rewards=R)
    states, rewards, dones, actions = ds_eval.get_trajectories(batch_size=100, timesteps=12, training=False)
    out = O.measure_prediction_mse(enc, tr, dec, rp, states, rewards, dones, actions, A)
    assert np.allclose(out["mse"], fx["eval_mse_pixel"], atol=1e-6)
    assert np.allclose(out["mse_std"], fx["eval_std_pixel"], atol=1e-6)
    assert np.allclose(out["rew"], fx["eval_mse_reward"], rtol=1e-4, atol=1e-5)
    assert np.allclose(out["rew_std"], fx["eval_std_reward"], rtol=1e-4, atol=1e-5)


def test_planner_rollout_vs_reference_run():
    """oracle.compute_rollout_reward vs the reference's own main.compute_rollout_reward run
    (tests/golden/planner.npz): every reward-head output along the 12-step beam and the best value."""
    fx = load("planner.npz")
    for A in (5, 4):
        tr = O.clone_state(WS.transition_state(200 + A, actions=A))
        rp = O.clone_state(WS.reward_state(210 + A, rewards=2))
        best, cum = O.compute_rollout_reward(torch.from_numpy(fx["A%d.z" % A]), tr, rp, A)
        want = fx["A%d.rewards" % A].sum(0)
        want[:, 0] *= 10.0                                              # main.py:484
        assert np.allclose(cum.numpy(), want, atol=2e-5, rtol=1e-5), np.abs(cum.numpy() - want).max()
        assert abs(float(best) - float(fx["A%d.best" % A])) < 2e-5


def test_excitation_backtrack_vs_reference_run():
    """oracle.excitation_backtrack vs the reference's own excitation_bptt.visualize_bptt run
    (tests/golden/bptt.npz): threshold step, rewards, d/da_t and d/dz_t of the arg-max reward pixel."""
    fx = load("bptt.npz")
    tr = O.clone_state(WS.transition_state(300, actions=4))
    rp = O.clone_state(WS.reward_state(301, rewards=2, gain=0.31))
    out = O.excitation_backtrack(tr, rp, torch.from_numpy(fx["z0"]), 4, [torch.from_numpy(n) for n in fx["noise"]])
    steps = int(fx["steps"])
    assert out is not None and out["t"] == steps - 1
    assert np.allclose(torch.stack(out["rewards"]).numpy(), fx["rewards"], atol=1e-5)
    for t in range(steps):
        assert np.array_equal(out["a"][t].numpy(), fx["a"][t])
        assert np.allclose(out["a_grad"][t].numpy(), fx["a_grad"][t], atol=1e-6, rtol=1e-3), t
        assert np.allclose(out["z_grad"][t].numpy(), fx["z_grad"][t], atol=1e-7, rtol=1e-3), t
    assert np.allclose(out["z0_grad"].numpy(), fx["z0_grad"], atol=1e-7, rtol=1e-3)
