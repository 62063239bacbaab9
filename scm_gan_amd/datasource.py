"""Batch interface of the hot path: the datasource contract of the reference.

Mirrors /root/reference/datasource.py:8-120: a datasource exposes
`binary_input_channels` (num actions), `scalar_output_channels` (num rewards),
`conv_input_channels`, `conv_output_channels`, `make_env`, `convert_frame` and

    get_trajectories(batch_size, timesteps, random_start=True, training=True)
        -> states  (B, T, C, H, W) float in [0, 1]
           rewards (B, T, R)       float
           dones   (B, T)          bool
           actions (B, T)          int

as host numpy arrays (reference envs/minipacman.py:122-164).  The reference's
simulators (gym, gym_minipacman, StarCraft II, ALE) are out of scope and not
available offline; SyntheticDatasource produces batches of the same shapes and
dtypes from a seeded numpy stream.
"""
import numpy as np


class Datasource(object):
    def convert_frame(self, state):
        # reference datasource.py:33-35
        return state, state


class SyntheticDatasource(Datasource):
    """i.i.d. sprite-like frames: pixels Bernoulli(density) in {0,1}, rewards
    U[0,1), uniform random actions; `done_every` > 0 marks dones[b, T//2] for
    every b % done_every == 0 so the active-mask path is exercised."""

    def __init__(self, seed=0, height=64, width=64, channels=3, num_actions=5, num_rewards=2,
                 density=0.1, done_every=0, dtype=np.float32):
        self.binary_input_channels = num_actions
        self.scalar_output_channels = num_rewards
        self.conv_input_channels = channels
        self.conv_output_channels = channels
        self.height, self.width = height, width
        self.density = density
        self.done_every = done_every
        self.dtype = dtype
        self._rng = {True: np.random.default_rng(seed), False: np.random.default_rng(seed + 104729)}

    def make_env(self, *args, **kwargs):
        raise NotImplementedError("SyntheticDatasource has no simulator (reference envs are out of scope)")

    def get_trajectories(self, batch_size=8, timesteps=10, random_start=True, training=True):
        rng = self._rng[bool(training)]
        B, T = batch_size, timesteps
        C, H, W = self.conv_input_channels, self.height, self.width
        states = (rng.random((B, T, C, H, W), dtype=np.float32) < self.density).astype(self.dtype)
        rewards = rng.random((B, T, self.scalar_output_channels), dtype=np.float32).astype(self.dtype)
        dones = np.zeros((B, T), dtype=bool)
        if self.done_every:
            dones[::self.done_every, T // 2] = True
        actions = rng.integers(0, self.binary_input_channels, size=(B, T))
        return states, rewards, dones, actions


_SYNTHETIC = {
    # name: (height, width, channels, actions, rewards) -- shapes of BASELINE.json configs
    "synthetic_minipacman": (64, 64, 3, 5, 2),
    "synthetic_betterpong": (84, 84, 3, 4, 1),
    "synthetic_sc2_star_intruders": (128, 128, 4, 4, 2),
    "synthetic": (64, 64, 3, 5, 2),
}

_REFERENCE_ENVS = ("sc2_star_intruders", "sc2_star_intruders_variant_a", "sc2_star_intruders_variant_b",
                   "sc2_star_intruders_variant_c", "pong", "gridworld", "gameoflife", "minipacman")


def allocate_datasource(datasource_name, seed=0):
    """Name -> datasource factory (reference datasource.py:8-26)."""
    if datasource_name in _SYNTHETIC:
        h, w, c, a, r = _SYNTHETIC[datasource_name]
        return SyntheticDatasource(seed=seed, height=h, width=w, channels=c, num_actions=a, num_rewards=r)
    if datasource_name in _REFERENCE_ENVS:
        raise ValueError(
            "datasource '%s' needs the reference's simulator stack (gym / gym_minipacman / sc2env), which is out "
            "of scope for this build; use 'synthetic_%s'-style names or pass any object with the "
            "get_trajectories contract" % (datasource_name, datasource_name))
    raise ValueError('Failed to find datasource with name {}'.format(datasource_name))
