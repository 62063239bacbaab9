"""The reference's three pure-numpy simulators as batch datasources.

Restated from the behaviour of /root/reference/envs/{gridworld,gameoflife,betterpong}.py and
pinned to them: tests/golden/envs.npz holds (state, action) -> (next state, reward, done, frame)
cases produced by the reference modules themselves (imported behind a stub of
gym.spaces.discrete.Discrete, tests/golden/make_golden.py gen_envs); tests/test_simulators_golden.py
requires exact equality, tests/test_host_logic.py adds rule tests.

Design differences, deliberate:
  * one vectorised state per batch (arrays over B) instead of B Python objects behind a
    thread pool (reference multi_env.py) -- the host side of an MI355X training loop must
    produce a batch in well under a step time;
  * a private seeded numpy Generator per datasource instead of the global numpy / random
    state, so runs are reproducible and data-parallel ranks draw different games: the random
    STREAM therefore differs from the reference's, the rules do not;
  * rewards come back as (B, T, R).  The reference's numpy envs wrap the reward list once
    more (`rewards = [rewards]`, e.g. gridworld.py:85) and return (1, T, B), which main.py:182
    then broadcasts against the (B, R) prediction into a (B, B) difference; that quirk is not
    reproduced (minipacman / sc2, the envs of the BASELINE configs, return (B, T, R)).

Episode conventions kept from the reference: the action recorded at step t is the one that
will be applied at step t+1 (the first applied action is drawn before the loop and never
returned); `done` is never raised; `training` and `random_start` are accepted and ignored.
"""
import numpy as np

from .datasource import Datasource


def _fill_rects(frames, channel, cx, cy, half_w, half_h):
    """frames[b, channel, top:bottom, left:right] = 1 with the reference's clipping
    (envs/gridworld.py:62-68): left/top clipped at 0, right/bottom at size-1, end-exclusive.
    The ends are Python slice stops: a negative one (a rectangle wholly left of / above the frame,
    not reachable in play) counts from the far edge, as in the reference."""
    B, _, H, W = frames.shape
    left = np.maximum(cx - half_w, 0)
    right = np.minimum(cx + half_w, W - 1)
    right = np.where(right < 0, np.maximum(right + W, 0), right)
    top = np.maximum(cy - half_h, 0)
    bottom = np.minimum(cy + half_h, H - 1)
    bottom = np.where(bottom < 0, np.maximum(bottom + H, 0), bottom)
    cols = np.arange(W)[None, :]
    rows = np.arange(H)[None, :]
    in_x = (cols >= left[:, None]) & (cols < right[:, None])          # (B, W)
    in_y = (rows >= top[:, None]) & (rows < bottom[:, None])          # (B, H)
    frames[:, channel] = np.maximum(frames[:, channel], (in_y[:, :, None] & in_x[:, None, :]).astype(frames.dtype))


class _BatchSimulator(Datasource):
    """Common trajectory loop (e.g. envs/gridworld.py:71-95)."""
    size = 64
    channels = 3
    num_actions = 4
    num_rewards = 1

    def __init__(self, seed=0, dtype=np.float32):
        self.binary_input_channels = self.num_actions
        self.scalar_output_channels = self.num_rewards
        self.conv_input_channels = self.channels
        self.conv_output_channels = self.channels
        self.height = self.width = self.size
        self.dtype = dtype
        self.rng = np.random.default_rng(seed)

    def make_env(self, *args, **kwargs):
        """A single game with the reset() / step(action) -> (state, reward, done, info) protocol that
        main.play() drives (reference main.py:331-371); extra arguments (screen_size=...) are ignored."""
        return SingleGame(self)

    # subclasses: reset(B) -> state dict, step(state, actions) -> rewards (B,), render(state) -> (B, C, H, W)
    def get_trajectories(self, batch_size=32, timesteps=10, policy='random', random_start=False, training=True):
        B, T = batch_size, timesteps
        st = self.reset(B)
        states = np.zeros((B, T, self.channels, self.size, self.size), dtype=self.dtype)
        rewards = np.zeros((B, T, self.num_rewards), dtype=self.dtype)
        dones = np.zeros((B, T), dtype=bool)
        actions = np.zeros((B, T), dtype=np.int64)
        act = self.rng.integers(0, self.num_actions, size=B)
        for t in range(T):
            rewards[:, t, 0] = self.step(st, act)
            states[:, t] = self.render(st)
            if policy == 'repeat':
                act = np.arange(B) % self.num_actions
            else:
                act = self.rng.integers(0, self.num_actions, size=B)
            actions[:, t] = act
        return states, rewards, dones, actions


class GridWorld(_BatchSimulator):
    """16x16 court, one 4x4-ish block moved by 4 buttons; +1 / -1 when it reaches the right / left
    wall (envs/gridworld.py:8-57)."""
    size, channels, num_actions, num_rewards = 16, 3, 4, 1
    RADIUS, MARGIN = 2, 2

    def reset(self, B):
        return {"x": self.rng.integers(self.MARGIN, self.size - self.MARGIN, size=B),
                "y": self.rng.integers(self.MARGIN, self.size - self.MARGIN, size=B)}

    def step(self, st, a):
        st["y"] = np.clip(st["y"] - 3 * (a == 0) + 3 * (a == 1), self.MARGIN, self.size - self.MARGIN)
        st["x"] = np.clip(st["x"] - 3 * (a == 2) + 3 * (a == 3), 0, self.size)
        return (st["x"] >= self.size).astype(np.float64) - (st["x"] <= 0).astype(np.float64)

    def render(self, st):
        f = np.zeros((len(st["x"]), self.channels, self.size, self.size), dtype=self.dtype)
        _fill_rects(f, 1, st["x"], st["y"], self.RADIUS, self.RADIUS)
        return f


class GameOfLife(_BatchSimulator):
    """Conway's Life on a 64x64 torus, one channel, a single no-op action, zero reward
    (envs/gameoflife.py:8-44)."""
    size, channels, num_actions, num_rewards = 64, 1, 1, 1

    def reset(self, B, p=0.5):
        return {"cells": self.rng.random((B, self.size, self.size)) > p}

    def step(self, st, a):
        c = st["cells"].astype(np.int8)
        n = sum(np.roll(np.roll(c, dy, 1), dx, 2) for dy in (-1, 0, 1) for dx in (-1, 0, 1)) - c
        st["cells"] = (n == 3) | (st["cells"] & (n == 2))
        return np.zeros(len(c))

    def render(self, st):
        return st["cells"][:, None].astype(self.dtype)


class Pong(_BatchSimulator):
    """Two-paddle pong on a 64x64 court (envs/betterpong.py:9-92): buttons 0/1 move the red (right)
    paddle up/down, 2/3 the blue (left) one; the ball bounces off paddles and the top/bottom walls;
    +1 when it leaves on the right, -1 on the left (it is then sent back)."""
    size, channels, num_actions, num_rewards = 64, 3, 4, 1
    PADDLE_W, PADDLE_H, RADIUS, MARGIN_Y, MARGIN_X = 1, 8, 2, 4, 5

    def reset(self, B):
        S = self.size
        speeds = np.array([-3, -2, 2, 3])
        return {"left_y": self.rng.integers(self.MARGIN_Y, S - self.MARGIN_Y, size=B),
                "right_y": self.rng.integers(self.MARGIN_Y, S - self.MARGIN_Y, size=B),
                "bx": self.rng.integers(self.MARGIN_X, S - self.MARGIN_X, size=B),
                "by": self.rng.integers(self.MARGIN_Y, S - self.MARGIN_Y, size=B),
                "vx": speeds[self.rng.integers(0, 4, size=B)], "vy": speeds[self.rng.integers(0, 4, size=B)]}

    def step(self, st, a):
        S = self.size
        st["right_y"] = np.clip(st["right_y"] - 3 * (a == 0) + 3 * (a == 1), 0, S)
        st["left_y"] = np.clip(st["left_y"] - 3 * (a == 2) + 3 * (a == 3), 0, S)
        bx, by, vx, vy = st["bx"] + st["vx"], st["by"] + st["vy"], st["vx"].copy(), st["vy"].copy()
        hit_r = S - self.MARGIN_X - self.RADIUS - self.PADDLE_W
        hit_l = self.MARGIN_X + self.RADIUS + self.PADDLE_W
        on_right = (bx >= hit_r) & (bx <= hit_r + self.RADIUS) & (vx > 0) & (np.abs(by - st["right_y"]) <= self.PADDLE_H)
        vx = np.where(on_right, -vx, vx)
        on_left = (bx >= hit_l - self.RADIUS) & (bx <= hit_l) & (vx < 0) & (np.abs(by - st["left_y"]) <= self.PADDLE_H)
        vx = np.where(on_left, -vx, vx)
        vy = np.where((by >= S - 2) & (vy > 0), -vy, vy)
        vy = np.where((by <= 2) & (vy < 0), -vy, vy)
        out_r = (bx >= S) & (vx > 0)
        vx = np.where(out_r, -vx, vx)
        out_l = (bx <= 0) & (vx < 0)
        vx = np.where(out_l, -vx, vx)
        st["bx"], st["by"], st["vx"], st["vy"] = bx, by, vx, vy
        return out_r.astype(np.float64) - out_l.astype(np.float64)

    def render(self, st):
        f = np.zeros((len(st["bx"]), self.channels, self.size, self.size), dtype=self.dtype)
        B = len(st["bx"])
        _fill_rects(f, 2, np.full(B, self.MARGIN_X), st["left_y"], self.PADDLE_W, self.PADDLE_H)
        _fill_rects(f, 0, np.full(B, self.size - self.MARGIN_X), st["right_y"], self.PADDLE_W, self.PADDLE_H)
        _fill_rects(f, 1, st["bx"], st["by"], self.RADIUS, self.RADIUS)
        return f



class SingleGame(object):
    """One instance (batch of 1) of a batch simulator behind the gym-style protocol."""

    def __init__(self, sim):
        self.sim = sim
        self.num_actions = sim.num_actions
        self.state = None
        self.reset()

    def reset(self):
        self.state = self.sim.reset(1)
        return self.sim.render(self.state)[0]

    def step(self, action):
        reward = float(self.sim.step(self.state, np.array([int(action)]))[0])
        return self.sim.render(self.state)[0], reward, False, {}
