"""CPU oracle for the scm-gan frame-prediction hot path.  TEST INFRASTRUCTURE.

A restatement, in plain torch CPU functional ops, of the arithmetic the
reference performs on this path.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this file; the product
(scm_gan_amd/) never does and has no CPU fallback.

Pinned (tests/test_oracle_golden.py) against golden vectors produced by the
reference's OWN modules and its own main.train()/measure_prediction_mse(),
imported from /root/reference in the build container
(tests/golden/make_golden.py).  The reference has no tests or fixtures of its
own, so those vectors are the pin.

State is a flat dict {reference state_dict key: tensor}; every function cites
the reference lines it follows (paths relative to /root/reference).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

LRELU = 0.01  # F.leaky_relu default negative_slope


# ---------------------------------------------------------------- primitives
def conv3x3(x, w, b, circular):
    """3x3 stride-1 same-size conv. zero pad: models.py:129-134; circular pad of
    one pixel per side: models.py:51-56 under torch<=1.4 semantics."""
    if circular:
        return F.conv2d(F.pad(x, (1, 1, 1, 1), mode="circular"), w, b)
    return F.conv2d(x, w, b, padding=1)


def l2normalize(v, eps=1e-12):
    """spectral_normalization.py:10-11"""
    return v / (v.norm() + eps)


def sn_weight(state, prefix):
    """SpectralNorm._update_u_v, one power iteration, u/v updated in place
    (no graph), sigma and W = W_bar / sigma in the graph.
    spectral_normalization.py:23-35."""
    u = state[prefix + ".module.weight_u"]
    v = state[prefix + ".module.weight_v"]
    w = state[prefix + ".module.weight_bar"]
    h = w.shape[0]
    wm = w.detach().view(h, -1)
    # The reference assigns u.data / v.data: the SAME tensor objects get new
    # storage without a version bump.  Graphs built by earlier calls saved those
    # objects, so at backward time EVERY call's d(sigma)/dW = u v^T is evaluated
    # with the latest u, v (while each call's own sigma value is kept).  The
    # oracle reproduces that literally.
    v.data = l2normalize(torch.mv(wm.t(), u.data))
    u.data = l2normalize(torch.mv(wm, v.data))
    sigma = u.dot(w.view(h, -1).mv(v))
    return w / sigma.expand_as(w), sigma


def bernoulli_st(p, noise):
    """DifferentiableBernoulliSampler (models.py:29-40) with the uniform noise
    made explicit: forward (u < p), backward identity."""
    hard = (noise < p).to(p.dtype)
    return p + (hard - p).detach()


# -------------------------------------------------------------------- models
def encoder_forward(state, x):
    """models.Encoder.forward, models.py:139-157 (bn_conv1 is never applied)."""
    b, frames, ch, h, w = x.shape
    x = x.reshape(b, frames * ch, h, w)
    for name in ("conv1", "conv2", "conv3"):
        wn, _ = sn_weight(state, name)
        x = F.leaky_relu(conv3x3(x, wn, state[name + ".module.bias"], False), LRELU)
    x = conv3x3(x, state["conv4.weight"], state["conv4.bias"], False)
    return torch.sigmoid(x)


def transition_forward(state, z, a, training, noise=None, return_all=False, return_logits=False):
    """models.Transition.forward, models.py:59-119."""
    b, _, h, w = z.shape
    amap = a.unsqueeze(-1).unsqueeze(-1).repeat(1, 1, h, w)
    x = torch.cat([z, amap], dim=1)
    wn, _ = sn_weight(state, "conv1")
    skip1 = F.leaky_relu(conv3x3(x, wn, state["conv1.module.bias"], True), LRELU)
    wn, _ = sn_weight(state, "conv2")
    skip2 = F.leaky_relu(conv3x3(skip1, wn, state["conv2.module.bias"], True), LRELU)
    wn, _ = sn_weight(state, "conv3")
    out3 = F.leaky_relu(conv3x3(skip2, wn, state["conv3.module.bias"], True), LRELU)
    wn, _ = sn_weight(state, "conv4")
    out4 = F.leaky_relu(conv3x3(out3, wn, state["conv4.module.bias"], True), LRELU)
    wn, _ = sn_weight(state, "conv5")
    out5 = F.leaky_relu(conv3x3(torch.cat([out4, skip2], 1), wn, state["conv5.module.bias"], True), LRELU)
    logits = conv3x3(torch.cat([out5, skip1], 1), state["conv6.weight"], state["conv6.bias"], True)
    p = torch.sigmoid(logits)
    if training:
        znext = bernoulli_st(p, noise)          # models.py:107-108
    else:
        znext = (p > 0.5).to(p.dtype)           # models.py:110-113
    if return_all:
        return (skip1, skip2, out3, out4, out5, znext) + ((logits,) if return_logits else ())
    if return_logits:
        return znext, logits
    return znext


def decoder_forward(state, z, visualize=False):
    """models.Decoder.forward, models.py:270-291."""
    b, latent, h, w = z.shape
    x = F.leaky_relu(F.conv_transpose2d(z, state["conv1.weight"], state["conv1.bias"], padding=1), LRELU)
    x = F.conv_transpose2d(x, state["conv2.weight"], state["conv2.bias"], padding=1)
    color = x.shape[1] // latent
    x = x.view(b, latent, color, h, w)
    out = torch.sum(x, dim=1)
    if visualize:
        return out, x[0]
    return out


def reward_forward(state, z, visualize=False):
    """models.RewardPredictor.forward, models.py:235-250."""
    x = F.leaky_relu(F.conv2d(z, state["conv1.weight"], state["conv1.bias"]), LRELU)
    x = F.conv2d(x, state["conv2.weight"], state["conv2.bias"], stride=2)
    b, ch, h, w = x.shape
    x = torch.softmax(x.view(b, 3, ch // 3, h, w), dim=1)
    x = x[:, 0] - x[:, 2]
    if visualize:
        return x.sum(-1).sum(-1), x
    return x.sum(-1).sum(-1)


def discriminator_forward(state, x):
    """models.Discriminator.forward, models.py:208-223, with the documented
    deviation: a row-major flatten in front of fc1 (the reference applies
    Linear(1568, 1) to an un-flattened map and cannot run)."""
    for name in ("conv1", "conv2"):
        wn, _ = sn_weight(state, name)
        x = F.leaky_relu(F.conv2d(x, wn, state[name + ".module.bias"], stride=2), LRELU)
    x = F.leaky_relu(F.conv2d(x, state["conv3.weight"], state["conv3.bias"], stride=2), LRELU)
    x = F.linear(x.reshape(x.shape[0], -1), state["fc1.weight"], state["fc1.bias"])
    return F.leaky_relu(x, LRELU)


def coordconv_forward(state, x, **conv_kwargs):
    """coordconv.CoordConv2d.forward, coordconv.py:10-15 (square inputs)."""
    b, c, h, w = x.shape
    cx = torch.arange(-1.0, 1.0, 2 / w).unsqueeze(0).repeat(w, 1).unsqueeze(0).repeat(b, 1, 1).unsqueeze(1)
    cy = torch.arange(-1.0, 1.0, 2 / h).unsqueeze(1).repeat(1, h).unsqueeze(0).repeat(b, 1, 1).unsqueeze(1)
    return F.conv2d(torch.cat([x, cx, cy], dim=1), state["conv.weight"], state["conv.bias"], **conv_kwargs)


def _gru_step(x, h, w_ih, w_hh):
    """One step of nn.GRU(bias=False), gate order r, z, n (torch docs)."""
    gi = x @ w_ih.t()
    gh = h @ w_hh.t()
    ir, iz, inn = gi.chunk(3, dim=1)
    hr, hz, hn = gh.chunk(3, dim=1)
    r = torch.sigmoid(ir + hr)
    zg = torch.sigmoid(iz + hz)
    n = torch.tanh(inn + r * hn)
    return (1 - zg) * n + zg * h


def csrn_forward(state, x):
    """spatial_recurrent.CSRN.forward, spatial_recurrent.py:46-119, including
    the :110 behaviour: the right sweep writes context_left (overwriting the
    left sweep) and context_right stays zero."""
    b, ch, h, w = x.shape
    ctx = {k: torch.zeros(b, ch, h, w) for k in ("above", "below", "left", "right")}

    def sweep(direction, order, along_rows, dest):
        n_tok = w if along_rows else h
        hstate = torch.zeros(b * n_tok, ch)
        for i in order:
            line = x[:, :, i, :] if along_rows else x[:, :, :, i]      # (b, ch, n_tok)
            tok = line.permute(0, 2, 1).reshape(b * n_tok, ch)
            out = _gru_step(tok, hstate, state["rnn_%s.weight_ih_l0" % direction],
                            state["rnn_%s.weight_hh_l0" % direction])
            conv_in = out.view(b, n_tok, ch).permute(0, 2, 1)
            if along_rows:
                dest[:, :, i, :] = conv_in
            else:
                dest[:, :, :, i] = conv_in
            conv_out = torch.tanh(F.conv1d(conv_in, state["conv_%s.weight" % direction],
                                           state["conv_%s.bias" % direction], padding=1))
            hstate = conv_out.permute(0, 2, 1).reshape(b * n_tok, ch)

    sweep("down", range(h), True, ctx["above"])
    sweep("up", reversed(range(h)), True, ctx["below"])
    sweep("left", range(w), False, ctx["left"])
    sweep("right", reversed(range(w)), False, ctx["left"])   # spatial_recurrent.py:110
    cmap = torch.cat((ctx["above"], ctx["below"], ctx["left"], ctx["right"]), dim=1)
    return F.conv2d(cmap, state["conv_combine.weight"], state["conv_combine.bias"])


# ------------------------------------------------------------- train / eval
def onehot(idx, num_actions):
    """main.py:447-452"""
    return torch.eye(num_actions)[torch.as_tensor(np.asarray(idx), dtype=torch.long)]


def decoder_pixel_loss(target, predicted):
    """main.py:310-312 (F.binary_cross_entropy clamps each log at -100)."""
    return F.binary_cross_entropy(predicted, target, reduction="none").mean(-1).mean(-1).mean(-1)


class Adam(object):
    """torch.optim.Adam defaults (main.py:125-129): betas (0.9, 0.999), eps 1e-8,
    no weight decay; parameters whose grad is None are skipped."""

    def __init__(self, params, lr):
        self.params = list(params)
        self.lr = lr
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.t = [0 for _ in self.params]

    def step(self):
        b1, b2, eps = 0.9, 0.999, 1e-8
        with torch.no_grad():
            for i, p in enumerate(self.params):
                if p.grad is None:
                    continue
                g = p.grad
                self.t[i] += 1
                t = self.t[i]
                self.m[i].mul_(b1).add_(g, alpha=1 - b1)
                self.v[i].mul_(b2).addcmul_(g, g, value=1 - b2)
                denom = (self.v[i].sqrt() / math.sqrt(1 - b2 ** t)).add_(eps)
                p.addcdiv_(self.m[i], denom, value=-self.lr / (1 - b1 ** t))


def trainable(state):
    """Parameters that receive gradients: everything floating point except
    SpectralNorm u/v (requires_grad=False, spectral_normalization.py:53-54) and
    BatchNorm running statistics."""
    out = []
    for k, v in state.items():
        if not v.dtype.is_floating_point:
            continue
        if k.endswith("weight_u") or k.endswith("weight_v") or "running_" in k:
            continue
        out.append(k)
    return out


class Trainer(object):
    """The per-iteration body of main.train(), main.py:143-297, for the default
    flag set (no truncate-bptt / latent overshooting / CF losses)."""

    def __init__(self, enc, tr, dec, rp, num_actions, lr=1e-4, reward_coef=1e-3):
        self.nets = {"enc": enc, "tr": tr, "dec": dec, "rp": rp}
        self.num_actions = num_actions
        self.reward_coef = reward_coef
        for st in self.nets.values():
            for k in trainable(st):
                st[k].requires_grad_(True)
        # bn_conv1.weight/bias are parameters of the reference Encoder but never
        # get a gradient; Adam skips them.
        self.opt = {n: Adam([st[k] for k in trainable(st)], lr) for n, st in self.nets.items()}

    def step(self, states, rewards, dones, actions, theta, noises, finetune_reward=False):
        """states (B,T,C,H,W), rewards (B,T,R), dones (B,T), actions (B,T) int;
        noises: list of (B,L,H,W) uniforms, one per transition call."""
        enc, tr, dec, rp = (self.nets[k] for k in ("enc", "tr", "dec", "rp"))
        for st in self.nets.values():
            for k in trainable(st):
                st[k].grad = None
        states = torch.as_tensor(states, dtype=torch.float32)
        rewards = torch.as_tensor(rewards, dtype=torch.float32)
        dones = torch.as_tensor(np.asarray(dones).astype(int), dtype=torch.float32)
        b, horizon = states.shape[0], states.shape[1]
        z = encoder_forward(enc, states[:, 0:3])                         # main.py:162
        active = torch.ones(b)
        loss = 0
        logs = []
        for t in range(1, horizon - 1):                                    # main.py:177
            active = active * (1 - dones[:, t])
            expected_reward = reward_forward(rp, z)
            rd = torch.mean(torch.mean((expected_reward - rewards[:, t]) ** 2, dim=1) * active)
            logs.append(("Rd Loss t={}".format(t), float(rd.detach())))
            loss = loss + theta * self.reward_coef * rd                    # main.py:185
            predicted = torch.sigmoid(decoder_forward(dec, z))
            rec = torch.mean(decoder_pixel_loss(states[:, t], predicted) * active)
            logs.append(("Reconstruction t={}".format(t), float(rec.detach())))
            loss = loss + rec                                              # main.py:197
            z = transition_forward(tr, z, onehot(actions[:, t], self.num_actions), True, noises[t - 1])
        loss.backward()                                                    # main.py:285
        for name in ("enc", "tr", "dec"):                                  # main.py:287-290
            for k in trainable(self.nets[name]):
                g = self.nets[name][k].grad
                if g is not None:
                    g.clamp_(-0.1, 0.1)
        self.opt["rp"].step()                                              # main.py:292
        if not finetune_reward:
            self.opt["enc"].step()
            self.opt["dec"].step()
            self.opt["tr"].step()
        return float(loss.detach()), logs


def measure_prediction_mse(enc, tr, dec, rp, states, rewards, dones, actions, num_actions, return_frames=False):
    """main.measure_prediction_mse, main.py:784-835 (numeric part), eval mode."""
    with torch.no_grad():
        states = torch.as_tensor(states, dtype=torch.float32)
        rewards = torch.as_tensor(rewards, dtype=torch.float32)
        dones = torch.as_tensor(np.asarray(dones).astype(int), dtype=torch.float32)
        b, timesteps = states.shape[0], states.shape[1]
        z = encoder_forward(enc, states[:, :3])
        z = transition_forward(tr, z, onehot(actions[:, 1], num_actions), False)
        out = {"mse": [], "mse_std": [], "rew": [], "rew_std": [], "frames": [], "z": []}
        active = torch.ones(b)
        for t in range(2, timesteps):
            active = active * (1 - dones[:, t])
            if active.sum() == 0:
                break
            predicted = torch.sigmoid(decoder_forward(dec, z))
            diffs = active * ((states[:, t] - predicted) ** 2).mean(dim=-1).mean(dim=-1).mean(dim=-1)
            out["mse"].append(float(torch.mean(diffs) * b / torch.sum(active)))
            out["mse_std"].append(float(torch.std(diffs) * b / torch.sum(active)))
            r_diffs = active * (rewards[:, t].sum(-1) - reward_forward(rp, z).sum(-1)) ** 2
            out["rew"].append(float(torch.mean(r_diffs) * b / torch.sum(active)))
            out["rew_std"].append(float(torch.std(r_diffs) * b / torch.sum(active)))
            if return_frames:
                out["frames"].append(predicted.clone())
                out["z"].append(z.clone())
            z = transition_forward(tr, z, onehot(actions[:, t], num_actions), False)
        return out


def compute_rollout_reward(z, tr, rp, num_actions, lookahead=2, rollout_depth=12, negative_positive_tradeoff=10.0):
    """main.compute_rollout_reward, main.py:455-489 (rollout_policy='noop'), eval mode."""
    with torch.no_grad():
        width = num_actions ** lookahead
        z = z.repeat(width, 1, 1, 1)
        actions = np.array([[i, j] + [0] * (rollout_depth - lookahead) for i in range(num_actions)
                            for j in range(num_actions)])
        cum = reward_forward(rp, z)
        for t in range(rollout_depth):
            z = transition_forward(tr, z, onehot(actions[:, t], num_actions), False)
            cum = cum + reward_forward(rp, z)
        cum[:, 0] *= negative_positive_tradeoff
        return cum.sum(dim=1).max(dim=0)[0], cum


def excitation_backtrack(tr, rp, z0, num_actions, noises, steps=30, threshold=0.8, first_action=1, later_action=3):
    """Numeric core of excitation_bptt.visualize_bptt, excitation_bptt.py:6-53: train-mode rollout
    (straight-through sampler, explicit noise) keeping gradients on every z_t and one-hot a_t; once
    |sum r| > threshold, backward from the arg-max pixel of the reward map (:36-37).
    Returns None or {'t', 'rewards', 'a', 'a_grad', 'z_grad', 'z0_grad'}."""
    z = z0.clone().requires_grad_(True)
    z_start = z
    acts, zees, rewards = [], [], []
    for t in range(steps):
        a = onehot(first_action if t == 0 else later_action, num_actions).unsqueeze(0).clone().requires_grad_(True)
        acts.append(a)
        z = transition_forward(tr, z, a, True, noises[t])
        z.retain_grad()
        zees.append(z)
        r, rmap = reward_forward(rp, z, visualize=True)
        rewards.append(r.detach().clone())
        if float(r.detach().sum().abs()) > threshold:
            peak = (rmap.abs() == rmap.abs().max()).to(rmap.dtype)
            (rmap * peak).sum().backward()
            return {"t": t, "rewards": rewards, "a": [x.detach() for x in acts], "a_grad": [x.grad for x in acts],
                    "z_grad": [x.grad for x in zees], "z0_grad": z_start.grad}
    return None


def clone_state(np_state):
    """numpy state dict (tests/golden/weights.py) -> fresh torch state."""
    return {k: torch.from_numpy(np.array(v)).clone() for k, v in np_state.items()}
