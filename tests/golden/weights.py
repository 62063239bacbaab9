"""Deterministic, platform-independent parameter sets for the parity fixtures.

Both the golden-vector generator (tests/golden/make_golden.py, which loads
them into the reference's own modules) and the tests (which load them into the
oracle and the HIP-backed modules) build their state_dicts here, so the
fixture files only have to hold inputs and expected outputs, not the 1.2 M
weights.  Key names and shapes follow the reference state_dicts
("State that must be reproduced exactly").
Values are drawn from numpy's PCG64 stream, with the fan-in scaling of torch's
default Conv2d init so activations stay in a realistic range.
"""
import numpy as np


def _conv(rng, cout, cin, k=3, gain=1.0):
    bound = gain / np.sqrt(cin * k * k)
    w = rng.uniform(-bound, bound, size=(cout, cin, k, k)).astype(np.float32)
    b = rng.uniform(-bound, bound, size=(cout,)).astype(np.float32)
    return w, b


def _unit(rng, n):
    v = rng.standard_normal(n).astype(np.float32)
    return (v / (np.linalg.norm(v) + 1e-12)).astype(np.float32)


def _sn(state, prefix, rng, cout, cin, gain=1.0):
    w, b = _conv(rng, cout, cin, gain=gain)
    state[prefix + ".module.bias"] = b
    state[prefix + ".module.weight_u"] = _unit(rng, cout)
    state[prefix + ".module.weight_v"] = _unit(rng, cin * 9)
    state[prefix + ".module.weight_bar"] = w


def _plain(state, prefix, rng, cout, cin, gain=1.0):
    w, b = _conv(rng, cout, cin, gain=gain)
    state[prefix + ".weight"] = w
    state[prefix + ".bias"] = b


def encoder_state(seed, latent=16, color=3, gain=3.0):
    rng = np.random.default_rng(seed)
    s = {}
    _sn(s, "conv1", rng, 128, color * 3, gain)
    s["bn_conv1.weight"] = np.ones(128, np.float32)
    s["bn_conv1.bias"] = np.zeros(128, np.float32)
    s["bn_conv1.running_mean"] = np.zeros(128, np.float32)
    s["bn_conv1.running_var"] = np.ones(128, np.float32)
    s["bn_conv1.num_batches_tracked"] = np.array(0, np.int64)
    _sn(s, "conv2", rng, 128, 128, gain)
    _sn(s, "conv3", rng, 128, 128, gain)
    _plain(s, "conv4", rng, latent, 128, gain)
    return s


def transition_state(seed, latent=16, actions=5, gain=3.0):
    rng = np.random.default_rng(seed)
    s = {}
    _sn(s, "conv1", rng, 128, latent + actions, gain)
    _sn(s, "conv2", rng, 128, 128, gain)
    _sn(s, "conv3", rng, 128, 128, gain)
    _sn(s, "conv4", rng, 128, 128, gain)
    _sn(s, "conv5", rng, 128, 256, gain)
    _plain(s, "conv6", rng, latent, 256, gain)
    return s


def decoder_state(seed, latent=16, color=3, gain=1.5):
    rng = np.random.default_rng(seed)
    s = {}
    # ConvTranspose2d weights are (Cin, Cout, 3, 3)
    w, b = _conv(rng, latent, latent * 4, gain=gain)
    s["conv1.weight"] = w
    s["conv1.bias"] = rng.uniform(-0.1, 0.1, size=(latent * 4,)).astype(np.float32)
    w, b = _conv(rng, latent * 4, latent * color, gain=gain)
    s["conv2.weight"] = w
    s["conv2.bias"] = rng.uniform(-0.1, 0.1, size=(latent * color,)).astype(np.float32)
    return s


def reward_state(seed, latent=16, rewards=2, gain=1.5):
    rng = np.random.default_rng(seed)
    s = {}
    _plain(s, "conv1", rng, 32, latent, gain)
    _plain(s, "conv2", rng, rewards * 3, 32, gain)
    return s


def discriminator_state(seed, gain=2.0):
    rng = np.random.default_rng(seed)
    s = {}
    _sn(s, "conv1", rng, 32, 3, gain)
    _sn(s, "conv2", rng, 32, 32, gain)
    _plain(s, "conv3", rng, 32, 32, gain)
    bound = 1.0 / np.sqrt(1568)
    s["fc1.weight"] = rng.uniform(-bound, bound, size=(1, 1568)).astype(np.float32)
    s["fc1.bias"] = rng.uniform(-bound, bound, size=(1,)).astype(np.float32)
    return s


def coordconv_state(seed, cin, cout):
    rng = np.random.default_rng(seed)
    s = {}
    _plain(s, "conv", rng, cout, cin + 2)
    return s


def csrn_state(seed, ch, scale=0.5):
    """CSRN parameters. The reference initialises conv/GRU weights N(0, ch)
    (spatial_recurrent.py:11-18), which saturates every tanh/sigmoid; `scale`
    picks the standard deviation so both regimes can be pinned."""
    rng = np.random.default_rng(seed)
    s = {}
    for d in ("down", "up", "left", "right"):
        s["conv_%s.weight" % d] = (rng.standard_normal((ch, ch, 3)) * scale).astype(np.float32)
        s["conv_%s.bias" % d] = rng.uniform(-0.2, 0.2, size=(ch,)).astype(np.float32)
        s["rnn_%s.weight_ih_l0" % d] = (rng.standard_normal((3 * ch, ch)) * scale).astype(np.float32)
        s["rnn_%s.weight_hh_l0" % d] = (rng.standard_normal((3 * ch, ch)) * scale).astype(np.float32)
    bound = 1.0 / np.sqrt(4 * ch)
    s["conv_combine.weight"] = rng.uniform(-bound, bound, size=(ch, 4 * ch, 1, 1)).astype(np.float32)
    s["conv_combine.bias"] = rng.uniform(-bound, bound, size=(ch,)).astype(np.float32)
    return s


def to_torch(state):
    import torch
    return {k: torch.from_numpy(np.array(v)) for k, v in state.items()}


def digest(t, n=192):
    """Compact fingerprint of a large tensor: a strided sample + moments."""
    a = np.asarray(t, dtype=np.float64).ravel()
    stride = max(1, a.size // n)
    return np.concatenate([[a.sum(), np.abs(a).sum(), np.sqrt((a * a).sum())], a[::stride][:n]])
