"""CPU: scm_gan_amd/simulators.py against the reference's own pure-numpy games.

tests/golden/envs.npz was produced by tests/golden/make_golden.py (gen_envs) from
/root/reference/envs/{gridworld,gameoflife,betterpong}.py imported behind a stub of
gym.spaces.discrete.Discrete (all they use of gym) and np.float: seeded random
(state, action) cases -> the reference's next state, reward, done and rendered frame.
Integer / boolean work: every comparison is exact."""
import numpy as np

from helpers import load


def _bits(fx, key, shape=None):
    shape = tuple(fx[key + "_shape"]) if shape is None else shape
    n = int(np.prod(shape))
    return np.unpackbits(fx[key])[:n].reshape(shape).astype(bool)


def test_gridworld_step_rule_and_render():
    from scm_gan_amd.simulators import GridWorld
    fx = load("envs.npz")
    sim = GridWorld(seed=0)
    st = {"x": fx["grid.x"].copy(), "y": fx["grid.y"].copy()}
    rew = sim.step(st, fx["grid.a"])
    assert np.array_equal(np.stack([st["x"], st["y"]], 1), fx["grid.next"])
    assert np.array_equal(rew, fx["grid.reward"])
    assert np.count_nonzero(fx["grid.reward"]) >= 8                   # walls were reached in the sample
    assert np.array_equal(sim.render(st) > 0.5, _bits(fx, "grid.frames"))


def test_gameoflife_step_rule():
    from scm_gan_amd.simulators import GameOfLife
    fx = load("envs.npz")
    boards = _bits(fx, "life.boards", (6, 64, 64))
    after = _bits(fx, "life.after", (6, 3, 64, 64))
    sim = GameOfLife(seed=0)
    st = {"cells": boards.copy()}
    for k in range(3):
        rew = sim.step(st, np.zeros(6, np.int64))
        assert np.array_equal(st["cells"], after[:, k]), k
        assert not rew.any()
        assert np.array_equal(sim.render(st)[:, 0] > 0.5, after[:, k])


def test_pong_step_rule_and_render():
    from scm_gan_amd.simulators import Pong
    fx = load("envs.npz")
    sim = Pong(seed=0)
    st = {"left_y": fx["pong.left_y"].copy(), "right_y": fx["pong.right_y"].copy(), "bx": fx["pong.bx"].copy(),
          "by": fx["pong.by"].copy(), "vx": fx["pong.vx"].copy(), "vy": fx["pong.vy"].copy()}
    rew = sim.step(st, fx["pong.a"])
    got = np.stack([st[k] for k in ("left_y", "right_y", "bx", "by", "vx", "vy")], 1)
    bad = np.nonzero((got != fx["pong.next"]).any(1))[0]
    assert bad.size == 0, (bad[:5], got[bad[:5]], fx["pong.next"][bad[:5]])
    assert np.array_equal(rew, fx["pong.reward"])
    # the sample exercises scoring on both sides and paddle / wall bounces
    assert (fx["pong.reward"] > 0).any() and (fx["pong.reward"] < 0).any()
    assert np.sum(np.sign(fx["pong.next"][:, 4]) != np.sign(fx["pong.vx"])) > 40
    frames = _bits(fx, "pong.frames")
    sub = {k: v[:frames.shape[0]] for k, v in st.items()}
    assert np.array_equal(sim.render(sub) > 0.5, frames)


def _grid_positions(frames):
    """Ball centre (x, y) from gridworld frames (B, T, 3, 16, 16), channel 1; None where the rectangle is
    clipped by a wall (the centre is then ambiguous)."""
    B, T = frames.shape[:2]
    pos = np.full((B, T, 2), -1)
    for b in range(B):
        for t in range(T):
            ys, xs = np.nonzero(frames[b, t, 1])
            if len(xs) and xs.max() - xs.min() == 3 and ys.max() - ys.min() == 3 and xs.max() < 15 and ys.max() < 15:
                pos[b, t] = (xs.min() + 2, ys.min() + 2)
    return pos


def test_trajectory_conventions_match_the_reference():
    """get_trajectories conventions read off the reference's own gridworld batch (np.random.seed(77), B=6,
    T=9) and required of ours: frames in {0,1}, dones never raised, and the action recorded at step t is the
    one applied between frame t and frame t+1 (envs/gridworld.py:71-95).  Ours returns rewards as
    (B, T, R); the reference's (1, T, B) wrapping (gridworld.py:85, `rewards = [rewards]`) is the documented
    deviation (simulators.py)."""
    from scm_gan_amd.simulators import GridWorld
    fx = load("envs.npz")
    ref_states = _bits(fx, "grid.traj_states")
    ref_actions = fx["grid.traj_actions"]
    assert fx["grid.traj_rewards"].shape == (1, 9, 6) and not fx["grid.traj_dones"].any()
    ours = GridWorld(seed=3).get_trajectories(batch_size=6, timesteps=9)
    assert ours[1].shape == (6, 9, 1) and not ours[2].any()
    checked = [0, 0]
    for which, (frames, actions) in enumerate(((ref_states, ref_actions), (ours[0] > 0.5, ours[3]))):
        pos = _grid_positions(frames)
        for b in range(6):
            for t in range(8):
                if (pos[b, t] < 0).any() or (pos[b, t + 1] < 0).any():
                    continue
                x, y = pos[b, t]
                a = actions[b, t]
                y2 = max(2, min(y - 3 * (a == 0) + 3 * (a == 1), 14))
                x2 = max(0, min(x - 3 * (a == 2) + 3 * (a == 3), 16))
                assert (x2, y2) == tuple(pos[b, t + 1]), (which, b, t, (x, y), a, tuple(pos[b, t + 1]))
                checked[which] += 1
    assert min(checked) >= 20
