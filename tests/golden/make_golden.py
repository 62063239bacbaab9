#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE's own code.

Runs only in the build container (needs /root/reference, read-only); nothing at
test time imports it.  It imports the reference's models.py,
spectral_normalization.py, coordconv.py, spatial_recurrent.py and main.py
unmodified, behind an in-process shim:

  1. stub modules `imutil` / `logutil` (plotting + logging only; absent here)
     and a stub `datasource` module whose allocate_datasource() returns our
     synthetic datasource (the reference's envs need gym/gym_minipacman/sc2env);
  2. `.cuda()` on tensors and modules becomes the identity (the reference
     hard-codes it; this container has no GPU);
  3. Transition's Conv2d(padding=2, padding_mode='circular') is built with
     padding=1: torch <= 1.4, which the reference was written for, padded
     (p+1)//2 = 1 pixel per side for that call;
  4. torch.bernoulli(p) is replaced by (u < p) with u drawn from a recorded
     numpy stream, so the sampled latents can be reproduced on the device;
  5. for the pure-numpy games (envs/gridworld.py, gameoflife.py, betterpong.py): a stub
     `gym.spaces.discrete.Discrete` (the only thing they use from gym) and
     `np.float = float` (removed from numpy 2.x; gameoflife.py:44 calls astype(np.float));
  6. tensor.type(torch.cuda.FloatTensor) -> float32 on the CPU (excitation_bptt.py:36),
     the same hard-coded-device issue as `.cuda()`.

Parameters come from tests/golden/weights.py (numpy PCG64), loaded into the
reference modules with load_state_dict, so fixtures hold inputs/outputs only.
"""
import argparse
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import weights as WS  # noqa: E402


# --------------------------------------------------------------------- shim
class _TimeSeries:
    """Stand-in for logutil.TimeSeries: records what train() collects."""
    log = []

    def __init__(self, *a, **k):
        pass

    def collect(self, name, value):
        _TimeSeries.log.append((name, float(value)))

    def print_every(self, n):
        pass

    def __str__(self):
        return "TimeSeries(stub)"


def install_shim(argv):
    imutil = types.ModuleType("imutil")
    imutil.show = lambda *a, **k: None
    imutil.Video = lambda *a, **k: None
    sys.modules["imutil"] = imutil
    logutil = types.ModuleType("logutil")
    logutil.TimeSeries = _TimeSeries
    logutil.sparkline = lambda *a, **k: ""
    sys.modules["logutil"] = logutil
    ds = types.ModuleType("datasource")
    ds.allocate_datasource = lambda name: None
    sys.modules["datasource"] = ds
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    _type = torch.Tensor.type

    def type_on_cpu(self, dtype=None, *a, **k):
        if dtype is torch.cuda.FloatTensor:
            dtype = torch.FloatTensor
        return _type(self, dtype, *a, **k)
    torch.Tensor.type = type_on_cpu
    # gym: the numpy games only need gym.spaces.discrete.Discrete(n).n
    gym = types.ModuleType("gym")
    spaces = types.ModuleType("gym.spaces")
    discrete = types.ModuleType("gym.spaces.discrete")

    class Discrete(object):
        def __init__(self, n):
            self.n = n
    discrete.Discrete = Discrete
    spaces.discrete, spaces.Discrete = discrete, Discrete
    gym.spaces = spaces
    sys.modules.update({"gym": gym, "gym.spaces": spaces, "gym.spaces.discrete": discrete})
    if not hasattr(np, "float"):
        np.float = float
    sys.argv = argv
    sys.path.insert(0, REF)


class circular_pad1:
    """Context: nn.Conv2d(padding=2, padding_mode='circular') -> padding=1."""

    def __enter__(self):
        self.orig = nn.Conv2d.__init__
        orig = self.orig

        def patched(mod, *a, **k):
            if k.get("padding_mode") == "circular" and k.get("padding") == 2:
                k["padding"] = 1
            orig(mod, *a, **k)
        nn.Conv2d.__init__ = patched

    def __exit__(self, *exc):
        nn.Conv2d.__init__ = self.orig


class recorded_bernoulli:
    """Context: torch.bernoulli(p) -> (u < p), u ~ U[0,1) from a numpy stream."""

    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.draws = []

    def __enter__(self):
        self.orig = torch.bernoulli

        def fake(p, *a, **k):
            u = torch.from_numpy(self.rng.random(tuple(p.shape), dtype=np.float32))
            self.draws.append(u.numpy().copy())
            return (u < p).to(p.dtype)
        torch.bernoulli = fake
        return self

    def __exit__(self, *exc):
        torch.bernoulli = self.orig


def npy(t):
    return t.detach().cpu().numpy().copy()


class Fixture(dict):
    """dict of arrays; tensors above `limit` elements are stored as a digest
    (weights.digest: 3 moments + a 4096-point strided sample) under key#digest."""
    limit = 70000

    def __setitem__(self, key, value):
        a = np.asarray(value)
        if a.size > self.limit and a.dtype.kind == "f":
            dict.__setitem__(self, key + "#digest", WS.digest(a, 4096))
            dict.__setitem__(self, key + "#shape", np.array(a.shape, np.int64))
        else:
            dict.__setitem__(self, key, a)


def grads_of(module):
    return {k: npy(p.grad) for k, p in module.named_parameters() if p.grad is not None}


def sn_state(module):
    return {k: npy(v) for k, v in module.state_dict().items() if k.endswith("weight_u") or k.endswith("weight_v")}


# ----------------------------------------------------------- module fixtures
KINK_BAND = 2e-6


def kink_ties(t):
    """Units of a leaky-relu OUTPUT map whose pre-activation lies within KINK_BAND of the kink (x = y for y >= 0,
    x = 100 y below): their gate -- and a 0.99 dy step in every gradient behind it -- is decided by the fp32
    summation order, not by the operator, so a gradient fixture must not contain any."""
    y = t.detach().double()
    return int((torch.where(y >= 0, y, 100.0 * y).abs() < KINK_BAND).sum())


def gen_modules(out, H, W, B, seed, with_grads=True, tag=""):
    """Gradient fixtures are drawn from the first input stream (1000 + seed + 7919 * attempt) whose hidden maps hold
    no unit on the leaky-relu kink (kink_ties): the strict gradient comparison then tests the operator in whatever
    summation order a backend uses."""
    for attempt in range(32):
        fx, ties = _gen_modules_once(H, W, B, seed, with_grads, 1000 + seed + 7919 * attempt)
        if not with_grads or ties == 0:
            break
        print("modules_%s: input stream %d has %d unit(s) on the leaky-relu kink; drawing another" % (tag, attempt, ties))
    assert not with_grads or ties == 0
    fx["kink_ties"] = np.array([ties, attempt], np.int64)
    np.savez_compressed(os.path.join(out, "modules_%s.npz" % tag), **fx)
    print("modules_%s.npz: %d arrays (input stream %d, %d kink ties)" % (tag, len(fx), attempt, ties))


def _gen_modules_once(H, W, B, seed, with_grads, data_seed):
    import models
    rng = np.random.default_rng(data_seed)
    A, R, C, L = 5, 2, 3, 16
    fx = Fixture()
    ties = 0

    # ---- Encoder (models.py:123-157)
    enc = models.Encoder(L, C)
    enc.load_state_dict(WS.to_torch(WS.encoder_state(seed)))
    enc.train()
    x = torch.from_numpy((rng.random((B, 3, C, H, W)) < 0.3).astype(np.float32))
    hidden = []
    hooks = [m.register_forward_hook(lambda m, i, o: hidden.append(torch.nn.functional.leaky_relu(o.detach(), 0.01)))
             for m in (enc.conv1, enc.conv2, enc.conv3)]
    z = enc(x)
    for h in hooks:
        h.remove()
    ties += sum(kink_ties(h) for h in hidden)
    fx["enc_x"] = npy(x)
    fx["enc_z"] = npy(z)
    if with_grads:
        gz = torch.from_numpy(rng.standard_normal(z.shape).astype(np.float32))
        (z * gz).sum().backward()
        fx["enc_gz"] = npy(gz)
        for k, g in grads_of(enc).items():
            fx["enc_grad." + k] = g
    for k, v in sn_state(enc).items():
        fx["enc_after1." + k] = v
    z2 = enc(x)  # second call: u/v advance again
    fx["enc_z_call2"] = npy(z2)

    # ---- Transition (models.py:43-119), train mode with recorded noise
    with circular_pad1():
        tr = models.Transition(L, A)
    tr.load_state_dict(WS.to_torch(WS.transition_state(seed + 1)))
    tr.train()
    zin = torch.from_numpy(rng.random((B, L, H, W)).astype(np.float32)).requires_grad_(True)
    act = np.zeros((B, A), np.float32)
    act[np.arange(B), rng.integers(0, A, B)] = 1.0
    a = torch.from_numpy(act).requires_grad_(True)
    logits = {}
    hook = tr.conv6.register_forward_hook(lambda m, i, o: logits.__setitem__("v", o.detach().clone()))
    with recorded_bernoulli(2000 + seed) as rb:
        outs = tr(zin, a, return_all=True)
    fx["tr_z"] = npy(zin)
    fx["tr_a"] = act
    fx["tr_noise"] = rb.draws[0]
    fx["tr_logits"] = npy(logits["v"])
    for name, o in zip(("skip1", "skip2", "out3", "out4", "out5", "znext"), outs):
        fx["tr_" + name] = npy(o)
    ties += sum(kink_ties(o) for o in outs[:5])
    if with_grads:
        gz = torch.from_numpy(rng.standard_normal(outs[-1].shape).astype(np.float32))
        (outs[-1] * gz).sum().backward()
        fx["tr_gz"] = npy(gz)
        fx["tr_grad_z"] = npy(zin.grad)
        fx["tr_grad_a"] = npy(a.grad)
        for k, g in grads_of(tr).items():
            fx["tr_grad." + k] = g
    for k, v in sn_state(tr).items():
        fx["tr_after1." + k] = v
    # eval mode, threshold (models.py:110-113); u/v advance a second time
    tr.eval()
    zb = torch.from_numpy((rng.random((B, L, H, W)) < 0.5).astype(np.float32))
    zn = tr(zb, torch.from_numpy(act))
    hook.remove()
    fx["tr_eval_z"] = npy(zb)
    fx["tr_eval_logits"] = npy(logits["v"])
    fx["tr_eval_znext"] = npy(zn)

    # ---- Decoder (models.py:253-291)
    dec = models.Decoder(L, C)
    dec.load_state_dict(WS.to_torch(WS.decoder_state(seed + 2)))
    zd = torch.from_numpy((rng.random((B, L, H, W)) < 0.5).astype(np.float32)).requires_grad_(True)
    y, vis = dec(zd, visualize=True)
    fx["dec_z"] = npy(zd)
    fx["dec_y"] = npy(y)
    fx["dec_vis"] = npy(vis)
    if with_grads:
        gy = torch.from_numpy(rng.standard_normal(y.shape).astype(np.float32))
        (y * gy).sum().backward()
        fx["dec_gy"] = npy(gy)
        fx["dec_grad_z"] = npy(zd.grad)
        for k, g in grads_of(dec).items():
            fx["dec_grad." + k] = g

    # ---- RewardPredictor (models.py:226-250)
    rp = models.RewardPredictor(L, R)
    rp.load_state_dict(WS.to_torch(WS.reward_state(seed + 3)))
    zr = torch.from_numpy((rng.random((B, L, H, W)) < 0.5).astype(np.float32)).requires_grad_(True)
    r, rmap = rp(zr, visualize=True)
    fx["rp_z"] = npy(zr)
    fx["rp_r"] = npy(r)
    fx["rp_map"] = npy(rmap)
    if with_grads:
        gr = torch.from_numpy(rng.standard_normal(r.shape).astype(np.float32))
        (r * gr).sum().backward()
        fx["rp_gr"] = npy(gr)
        fx["rp_grad_z"] = npy(zr.grad)
        for k, g in grads_of(rp).items():
            fx["rp_grad." + k] = g

    return fx, ties


def gen_operators(out, seed):
    """SpectralNorm call sequence, Discriminator (+ documented flatten),
    CoordConv2d, CSRN."""
    import models
    from spectral_normalization import SpectralNorm
    from coordconv import CoordConv2d
    from spatial_recurrent import CSRN
    rng = np.random.default_rng(3000 + seed)
    fx = Fixture()

    # ---- SpectralNorm (spectral_normalization.py:14-68): three calls
    st = WS.transition_state(seed)
    conv = nn.Conv2d(128, 128, 3, padding=1)
    sn = SpectralNorm(conv)
    sn.load_state_dict({"module." + k.split(".module.")[1]: torch.from_numpy(v)
                        for k, v in st.items() if k.startswith("conv2.module.")})
    x = torch.from_numpy(rng.standard_normal((1, 128, 6, 6)).astype(np.float32))
    for call in (1, 2, 3):
        y = sn(x)
        w_bar = conv.weight_bar.detach()
        sigma = float(conv.weight_u.detach().dot(w_bar.view(128, -1).mv(conv.weight_v.detach())))
        fx["sn_u%d" % call] = npy(conv.weight_u)
        fx["sn_v%d" % call] = npy(conv.weight_v)
        fx["sn_sigma%d" % call] = np.float64(sigma)
        fx["sn_y%d" % call] = npy(y)
    fx["sn_x"] = npy(x)
    gy = torch.from_numpy(rng.standard_normal(y.shape).astype(np.float32))
    (y * gy).sum().backward()
    fx["sn_gy"] = npy(gy)
    fx["sn_grad_wbar"] = npy(conv.weight_bar.grad)
    fx["sn_grad_bias"] = npy(conv.bias.grad)

    # ---- Discriminator (models.py:195-223). Its forward cannot run as written
    # (Linear applied to an un-flattened B x 32 x 7 x 7 map, models.py:205,221);
    # the documented deviation is a row-major flatten in front of fc1, applied
    # here by wrapping fc1 -- the reference forward itself runs unchanged.
    class FlattenThen(nn.Module):
        def __init__(self, lin):
            super().__init__()
            self.lin = lin

        def forward(self, t):
            return self.lin(t.view(t.shape[0], -1))

    disc = models.Discriminator()
    disc.load_state_dict(WS.to_torch(WS.discriminator_state(seed + 4)))
    disc.fc1 = FlattenThen(disc.fc1)
    xd = torch.from_numpy((rng.random((3, 3, 64, 64)) < 0.5).astype(np.float32)).requires_grad_(True)
    logit = disc(xd)
    fx["disc_x"] = npy(xd)
    fx["disc_logit"] = npy(logit)
    gl = torch.from_numpy(rng.standard_normal(logit.shape).astype(np.float32))
    (logit * gl).sum().backward()
    fx["disc_gl"] = npy(gl)
    fx["disc_grad_x"] = npy(xd.grad)
    for k, p in disc.named_parameters():
        if p.grad is not None:
            fx["disc_grad." + k.replace("fc1.lin.", "fc1.")] = npy(p.grad)
    for k, v in sn_state(disc).items():
        fx["disc_after1." + k] = v

    # ---- CoordConv2d (coordconv.py:5-15), square input
    cc = CoordConv2d(6 + 2, 10, 3, padding=1)
    cc.load_state_dict(WS.to_torch(WS.coordconv_state(seed + 5, 6, 10)))
    xc = torch.from_numpy(rng.standard_normal((2, 6, 12, 12)).astype(np.float32)).requires_grad_(True)
    yc = cc(xc)
    gc = torch.from_numpy(rng.standard_normal(yc.shape).astype(np.float32))
    (yc * gc).sum().backward()
    fx["cc_x"], fx["cc_y"], fx["cc_gy"], fx["cc_grad_x"] = npy(xc), npy(yc), npy(gc), npy(xc.grad)
    fx["cc_grad.conv.weight"] = npy(cc.conv.weight.grad)
    fx["cc_grad.conv.bias"] = npy(cc.conv.bias.grad)

    # ---- CSRN (spatial_recurrent.py:21-119), incl. the :110 overwrite
    for tag, scale in (("soft", 0.5), ("sat", 8.0)):
        ch = 8
        cs = CSRN(ch)
        cs.load_state_dict(WS.to_torch(WS.csrn_state(seed + 6, ch, scale)))
        xs = torch.from_numpy(rng.standard_normal((2, ch, 6, 6)).astype(np.float32)).requires_grad_(True)
        ys = cs(xs)
        gs = torch.from_numpy(rng.standard_normal(ys.shape).astype(np.float32))
        (ys * gs).sum().backward()
        fx["csrn_%s_x" % tag], fx["csrn_%s_y" % tag] = npy(xs), npy(ys)
        fx["csrn_%s_gy" % tag], fx["csrn_%s_grad_x" % tag] = npy(gs), npy(xs.grad)
        for k, p in cs.named_parameters():
            fx["csrn_%s_grad.%s" % (tag, k)] = npy(p.grad) if p.grad is not None else np.zeros(p.shape, np.float32)

    np.savez_compressed(os.path.join(out, "operators.npz"), **fx)
    print("operators.npz: %d arrays" % len(fx))


# ------------------------------------------------- train() and rollout fixtures
def gen_train(out, seed, H=16, W=16, B=4, horizon=5, iters=3):
    """Run the reference's own main.train() (main.py:106-299) for a few
    iterations on the synthetic datasource, then its measure_prediction_mse
    (main.py:784-855)."""
    import main as ref_main
    import models
    from scm_gan_amd.datasource import SyntheticDatasource
    A, R, C, L = 5, 2, 3, 16
    ds = SyntheticDatasource(seed=seed, height=H, width=W, channels=C, num_actions=A, num_rewards=R,
                             done_every=4)
    enc = models.Encoder(L, C)
    dec = models.Decoder(L, C)
    rp = models.RewardPredictor(L, R)
    disc = models.Discriminator()
    with circular_pad1():
        tr = models.Transition(L, A)
    enc.load_state_dict(WS.to_torch(WS.encoder_state(seed)))
    tr.load_state_dict(WS.to_torch(WS.transition_state(seed + 1)))
    dec.load_state_dict(WS.to_torch(WS.decoder_state(seed + 2)))
    rp.load_state_dict(WS.to_torch(WS.reward_state(seed + 3)))
    disc.load_state_dict(WS.to_torch(WS.discriminator_state(seed + 4)))

    # record the gradients every Adam instance sees on its first step
    grads_first = {}
    orig_step = torch.optim.Adam.step

    def step(self, *a, **k):
        for gi, g in enumerate(self.param_groups):
            for pi, p in enumerate(g["params"]):
                key = (id(self), pi)
                if p.grad is not None and key not in grads_first:
                    grads_first[key] = npy(p.grad)
        return orig_step(self, *a, **k)
    torch.optim.Adam.step = step
    _TimeSeries.log = []
    a = ref_main.args
    a.batch_size, a.train_iters, a.horizon_min, a.horizon_max = B, iters, horizon, horizon
    fx = {"cfg": np.array([B, H, W, horizon, iters, A, R, C, L, seed], np.int64)}
    with recorded_bernoulli(5000 + seed) as rb:
        ref_main.train(L, ds, A, R, enc, dec, rp, disc, tr)
    torch.optim.Adam.step = orig_step
    fx["train_noise_draws"] = np.int64(len(rb.draws))
    fx["train_loss_names"] = np.array([n for n, _ in _TimeSeries.log])
    fx["train_loss_values"] = np.array([v for _, v in _TimeSeries.log], np.float64)
    for name, m in (("enc", enc), ("tr", tr), ("dec", dec), ("rp", rp)):
        for k, v in m.state_dict().items():
            if v.dtype.is_floating_point:
                fx["final.%s.%s" % (name, k)] = WS.digest(npy(v))
    # first-step gradients, keyed by module/param order
    # (Adam instances were created enc, dec, trans, disc, pred: main.py:125-129)
    by_opt = {}
    for (oid, pi), g in grads_first.items():
        by_opt.setdefault(oid, {})[pi] = g
    opt_ids = sorted(by_opt.keys(), key=lambda o: list(grads_first.keys()).index((o, min(by_opt[o]))))
    # map parameter tensors back to names through shapes in module order
    for name, m in (("enc", enc), ("tr", tr), ("dec", dec), ("rp", rp)):
        plist = list(m.named_parameters())
        for oid in opt_ids:
            if len(by_opt[oid]) and all(pi < len(plist) for pi in by_opt[oid]) and \
                    all(plist[pi][1].shape == by_opt[oid][pi].shape for pi in by_opt[oid]) and \
                    ("claimed", oid) not in fx:
                # shapes match this module: claim it
                for pi, g in by_opt[oid].items():
                    fx["grad1.%s.%s" % (name, plist[pi][0])] = WS.digest(g)
                fx[("claimed", oid)] = True
                break
    fx = {k: v for k, v in fx.items() if not isinstance(k, tuple)}

    # ---- eval rollout with the trained weights: main.py:784-855
    captured = {}

    def capture(mse_losses, mse_stddevs, experiment_name, train_iter, **k):
        captured[experiment_name] = (list(mse_losses), list(mse_stddevs))
    ref_main.plot_error_graph = capture
    ref_main.test_mode([enc, dec, tr, disc, rp])
    ds_eval = SyntheticDatasource(seed=seed + 77, height=H, width=W, channels=C, num_actions=A, num_rewards=R,
                                  done_every=0)
    ref_main.measure_prediction_mse(ds_eval, enc, dec, tr, rp, train_iter=0, timesteps=12)
    fx["eval_mse_pixel"] = np.array(captured["pixel"][0], np.float64)
    fx["eval_std_pixel"] = np.array(captured["pixel"][1], np.float64)
    fx["eval_mse_reward"] = np.array(captured["reward"][0], np.float64)
    fx["eval_std_reward"] = np.array(captured["reward"][1], np.float64)
    np.savez_compressed(os.path.join(out, "train_rollout.npz"), **fx)
    print("train_rollout.npz: %d arrays, %d losses, %d noise draws" % (len(fx), len(_TimeSeries.log), len(rb.draws)))


def gen_train_flags(out, seed):
    """The optional loss terms of the reference's train() (main.py:192-193,217-234,
    242-283): one short run per flag set, same data / noise protocol as gen_train."""
    import main as ref_main
    import models
    from scm_gan_amd.datasource import SyntheticDatasource
    A, R, C, L = 5, 2, 3, 16
    H = W = 12
    B = 4
    fx = {}
    cases = {
        # --truncate-bptt cannot be pinned: the reference z.detach_() (main.py:193) raises on torch 2.x
        # ("saved tensor ... detached in-place"); the build implements it as an out-of-place detach.
        "lo": dict(iters=2, horizon=6, latent_overshooting=True),
        "cf": dict(iters=5, horizon=5, enable_disentanglement_loss=True, enable_action_control_loss=True,
                   counterfactual_horizon=3),
    }
    for name, cfg in cases.items():
        ds = SyntheticDatasource(seed=seed + 5, height=H, width=W, channels=C, num_actions=A, num_rewards=R,
                                 done_every=3)
        enc, dec, rp, disc = models.Encoder(L, C), models.Decoder(L, C), models.RewardPredictor(L, R), \
            models.Discriminator()
        with circular_pad1():
            tr = models.Transition(L, A)
        enc.load_state_dict(WS.to_torch(WS.encoder_state(seed + 30)))
        tr.load_state_dict(WS.to_torch(WS.transition_state(seed + 31)))
        dec.load_state_dict(WS.to_torch(WS.decoder_state(seed + 32)))
        rp.load_state_dict(WS.to_torch(WS.reward_state(seed + 33)))
        a = ref_main.args
        a.batch_size, a.train_iters, a.horizon_min, a.horizon_max = B, cfg["iters"], cfg["horizon"], cfg["horizon"]
        a.truncate_bptt = cfg.get("truncate_bptt", False)
        a.latent_overshooting = cfg.get("latent_overshooting", False)
        a.counterfactual_horizon = cfg.get("counterfactual_horizon", 1)
        # these two are module-level copies in the reference (main.py:57-58)
        ref_main.enable_cf_shuffle_loss = cfg.get("enable_disentanglement_loss", False)
        ref_main.enable_control_bias_loss = cfg.get("enable_action_control_loss", False)
        _TimeSeries.log = []
        np.random.seed(1234)
        with recorded_bernoulli(7000 + seed) as rb:
            ref_main.train(L, ds, A, R, enc, dec, rp, disc, tr)
        fx[name + ".cfg"] = np.array([B, H, W, cfg["horizon"], cfg["iters"], cfg.get("counterfactual_horizon", 1)], np.int64)
        fx[name + ".noise_shapes"] = np.array([d.shape[0] for d in rb.draws], np.int64)
        fx[name + ".loss_names"] = np.array([n for n, _ in _TimeSeries.log])
        fx[name + ".loss_values"] = np.array([v for _, v in _TimeSeries.log], np.float64)
        for mname, m in (("enc", enc), ("tr", tr), ("dec", dec), ("rp", rp)):
            for k, v in m.state_dict().items():
                if v.dtype.is_floating_point:
                    fx["%s.final.%s.%s" % (name, mname, k)] = WS.digest(npy(v))
        print(name, len(_TimeSeries.log), "losses,", len(rb.draws), "noise draws")
    a.truncate_bptt = a.latent_overshooting = False
    a.counterfactual_horizon = 1
    ref_main.enable_cf_shuffle_loss = ref_main.enable_control_bias_loss = False
    np.savez_compressed(os.path.join(out, "train_flags.npz"), **fx)


def gen_planner(out, seed):
    """The reference's own main.compute_rollout_reward (main.py:455-489), eval mode under no_grad as
    main.py:95 runs play(): best cumulative reward of the 25- (A=5) and 16-beam (A=4) rollouts, plus
    every reward-head output along the way (forward hook)."""
    import main as ref_main
    import models
    L, R = 16, 2
    fx = {}
    for A, size in ((5, 16), (4, 12)):
        with circular_pad1():
            tr = models.Transition(L, A)
        rp = models.RewardPredictor(L, R)
        tr.load_state_dict(WS.to_torch(WS.transition_state(seed + 200 + A, actions=A)))
        rp.load_state_dict(WS.to_torch(WS.reward_state(seed + 210 + A, rewards=R)))
        ref_main.test_mode([tr, rp])
        rng = np.random.default_rng(8000 + A)
        z = torch.from_numpy((rng.random((1, L, size, size)) < 0.5).astype(np.float32))
        seen = []
        hook = rp.register_forward_hook(lambda m, i, o: seen.append(npy(o)))
        with torch.no_grad():
            best = ref_main.compute_rollout_reward(z, tr, rp, A, 0, rollout_depth=12, rollout_policy='noop')
        hook.remove()
        fx["A%d.z" % A] = npy(z)
        fx["A%d.best" % A] = np.float64(float(best))
        fx["A%d.rewards" % A] = np.stack(seen)            # (13, A*A, R): reward head after 0..12 transitions
    np.savez_compressed(os.path.join(out, "planner.npz"), **fx)
    print("planner.npz:", {k: np.asarray(v).shape for k, v in fx.items()})


def gen_bptt(out, seed):
    """The numeric core of the reference's excitation_bptt.visualize_bptt (excitation_bptt.py:6-53) run as
    is (video writer stubbed): train-mode transition (straight-through sampler, recorded noise), backward
    from the arg-max pixel of the reward map once |r| > 0.8; recorded through a wrapper on
    transition.forward: every a_t, a_t.grad, z_t.grad."""
    import excitation_bptt as ref_bptt
    import models
    L, A, R, size = 16, 4, 2, 16            # visualize_bptt's onehot() is hard-wired to 4 actions
    with circular_pad1():
        tr = models.Transition(L, A)
    rp = models.RewardPredictor(L, R)
    dec = models.Decoder(L, 3)
    tr.load_state_dict(WS.to_torch(WS.transition_state(seed + 300, actions=A)))
    rp.load_state_dict(WS.to_torch(WS.reward_state(seed + 301, rewards=R, gain=0.31)))
    dec.load_state_dict(WS.to_torch(WS.decoder_state(seed + 302)))
    tr.train()
    rng = np.random.default_rng(8100)
    z0 = torch.from_numpy((rng.random((1, L, size, size)) < 0.5).astype(np.float32)).requires_grad_(True)
    calls = []
    orig_forward = tr.forward

    def recording_forward(z, a, *args, **kw):
        o = orig_forward(z, a, *args, **kw)
        calls.append((a, o))
        return o
    tr.forward = recording_forward

    class Vid(object):
        frames = 0

        def write_frame(self, *a, **k):
            Vid.frames += 1
    rewards_seen = []
    hook = rp.register_forward_hook(lambda m, i, o: rewards_seen.append(npy(o[0])))
    with recorded_bernoulli(8200 + seed) as rb:
        ref_bptt.visualize_bptt(z0, tr, rp, dec, lambda x: x, A, vid=Vid())
    hook.remove()
    steps = len(calls)
    assert 1 < steps < 30, "pick reward-head weights so the 0.8 threshold is crossed after a few steps (got %d)" % steps
    fx = {"z0": npy(z0), "steps": np.int64(steps), "noise": np.stack(rb.draws),
          "rewards": np.stack(rewards_seen),                          # (steps, 1, R)
          "a": np.stack([npy(a) for a, _ in calls]),
          "a_grad": np.stack([npy(a.grad) for a, _ in calls]),
          "z_grad": np.stack([npy(o.grad) for _, o in calls]),
          "z0_grad": npy(z0.grad)}
    np.savez_compressed(os.path.join(out, "bptt.npz"), **fx)
    print("bptt.npz: threshold crossed at step %d; |r| per step %s" % (steps - 1, np.abs(fx["rewards"].sum(-1)).ravel()))


def gen_envs(out, seed):
    """Step rules of the reference's pure-numpy games, from the reference modules themselves
    (envs/gridworld.py, gameoflife.py, betterpong.py behind the gym / np.float stubs): for seeded random
    (state, action) cases -> next state, reward, done, rendered frame; plus one seeded get_trajectories()
    call of gridworld (whose state is fully visible in its frames) for the batch conventions."""
    sys.path.insert(0, os.path.join(REF, "envs"))
    import gridworld
    import gameoflife
    import betterpong
    rng = np.random.default_rng(9000 + seed)
    fx = {}
    # ---- gridworld (envs/gridworld.py:18-57)
    n = 96
    env = gridworld.Env()
    x0, y0 = rng.integers(0, 17, n), rng.integers(2, 15, n)
    act = rng.integers(0, 4, n)
    nxt, rew, frames = [], [], []
    for i in range(n):
        env.ball_x, env.ball_y = int(x0[i]), int(y0[i])
        st, r, done, info = env.step(int(act[i]))
        assert done is False
        nxt.append((env.ball_x, env.ball_y))
        rew.append(r)
        frames.append(st)
    fx["grid.x"], fx["grid.y"], fx["grid.a"] = x0, y0, act
    fx["grid.next"] = np.array(nxt, np.int64)
    fx["grid.reward"] = np.array(rew, np.float64)
    fx["grid.frames"] = np.packbits(np.array(frames) > 0.5)
    fx["grid.frames_shape"] = np.array(np.array(frames).shape, np.int64)
    np.random.seed(77)
    s, r, d, a = gridworld.get_trajectories(batch_size=6, timesteps=9)
    fx["grid.traj_states"] = np.packbits(np.asarray(s) > 0.5)
    fx["grid.traj_states_shape"] = np.array(np.asarray(s).shape, np.int64)
    fx["grid.traj_rewards"] = np.asarray(r, np.float64)          # (1, T, B): the reference's reward layout quirk
    fx["grid.traj_dones"] = np.asarray(d)
    fx["grid.traj_actions"] = np.asarray(a, np.int64)
    # ---- game of life (envs/gameoflife.py:18-44): 6 boards, 3 consecutive steps each
    boards, after = [], []
    for i in range(6):
        env = gameoflife.Env()
        env.state = rng.random((64, 64)) > (0.3 + 0.1 * i)
        boards.append(env.state.copy())
        seq = []
        for _ in range(3):
            st, r, done, info = env.step(0)
            assert r == 0 and done is False and st.shape == (1, 64, 64)
            seq.append(st[0] > 0.5)
        after.append(seq)
    fx["life.boards"] = np.packbits(np.array(boards))
    fx["life.after"] = np.packbits(np.array(after))
    # ---- pong (envs/betterpong.py:24-92): random states incl. positions past the walls and at the paddles
    n = 512
    env = betterpong.BetterPongEnv()
    cols = {k: [] for k in ("left_y", "right_y", "bx", "by", "vx", "vy", "a")}
    nxt, rew, frames = [], [], []
    speeds = [-3, -2, 2, 3]
    for i in range(n):
        near_paddle = i % 3 == 0
        c = dict(left_y=int(rng.integers(0, 65)), right_y=int(rng.integers(0, 65)),
                 bx=int(rng.choice([5, 6, 7, 8, 9, 10, 54, 55, 56, 57, 58, 59])) if near_paddle else int(rng.integers(-4, 69)),
                 by=int(rng.integers(-2, 67)), vx=int(rng.choice(speeds)), vy=int(rng.choice(speeds)), a=int(rng.integers(0, 4)))
        if near_paddle and i % 2 == 0:      # make paddle hits likely
            c["by"] = int(np.clip((c["right_y"] if c["bx"] > 32 else c["left_y"]) + rng.integers(-9, 10), -2, 66))
        env.left_y, env.right_y, env.ball_x, env.ball_y = c["left_y"], c["right_y"], c["bx"], c["by"]
        env.ball_velocity_x, env.ball_velocity_y = c["vx"], c["vy"]
        st, r, done, info = env.step(c["a"])
        assert done is False
        for k in cols:
            cols[k].append(c[k])
        nxt.append((env.left_y, env.right_y, env.ball_x, env.ball_y, env.ball_velocity_x, env.ball_velocity_y))
        rew.append(r)
        if i < 96:
            frames.append(st)
    for k, v in cols.items():
        fx["pong." + k] = np.array(v, np.int64)
    fx["pong.next"] = np.array(nxt, np.int64)
    fx["pong.reward"] = np.array(rew, np.float64)
    fx["pong.frames"] = np.packbits(np.array(frames) > 0.5)
    fx["pong.frames_shape"] = np.array(np.array(frames).shape, np.int64)
    np.savez_compressed(os.path.join(out, "envs.npz"), **fx)
    print("envs.npz: gridworld %d cases (%d scored), life 6x3 boards, pong %d cases (%d scored, %d x-bounces)" % (
        len(x0), int(np.count_nonzero(fx["grid.reward"])), n, int(np.count_nonzero(fx["pong.reward"])),
        int(np.sum(np.sign(fx["pong.next"][:, 4]) != np.sign(fx["pong.vx"])))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=HERE)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    install_shim(["main.py", "--env", "synthetic"])
    torch.manual_seed(0)
    torch.set_num_threads(8)
    if a.only in ("", "modules"):
        gen_modules(a.out, 16, 16, 2, a.seed, with_grads=True, tag="16")
        gen_modules(a.out, 20, 12, 1, a.seed + 10, with_grads=True, tag="20x12")
        gen_modules(a.out, 64, 64, 1, a.seed + 20, with_grads=False, tag="64")
    if a.only in ("", "operators"):
        gen_operators(a.out, a.seed)
    if a.only in ("", "train"):
        gen_train(a.out, a.seed)
    if a.only in ("", "flags"):
        gen_train_flags(a.out, a.seed)
    if a.only in ("", "planner"):
        gen_planner(a.out, a.seed)
    if a.only in ("", "bptt"):
        gen_bptt(a.out, a.seed)
    if a.only in ("", "envs"):
        gen_envs(a.out, a.seed)


if __name__ == "__main__":
    main()
