"""CPU: host-side logic of the drop-in -- datasource contract, state_dict layout
(checkpoint interchange with the reference), schedules, logging stand-in."""
import numpy as np
import pytest
import torch

import weights as WS


def test_datasource_contract():
    from scm_gan_amd.datasource import SyntheticDatasource, allocate_datasource
    ds = SyntheticDatasource(seed=3, height=20, width=12, channels=3, num_actions=5, num_rewards=2, done_every=2)
    assert (ds.binary_input_channels, ds.scalar_output_channels) == (5, 2)
    assert (ds.conv_input_channels, ds.conv_output_channels) == (3, 3)
    s, r, d, a = ds.get_trajectories(4, 7)
    assert s.shape == (4, 7, 3, 20, 12) and r.shape == (4, 7, 2) and d.shape == (4, 7) and a.shape == (4, 7)
    assert d.dtype == bool and set(np.unique(s)) <= {0.0, 1.0} and a.min() >= 0 and a.max() < 5
    assert d[0, 3] and d[2, 3] and not d[1].any()
    # deterministic per seed, separate train / test streams
    s2 = SyntheticDatasource(seed=3, height=20, width=12, channels=3, done_every=2).get_trajectories(4, 7)[0]
    assert np.array_equal(s, s2)
    st = ds.get_trajectories(4, 7, training=False)[0]
    assert not np.array_equal(st, ds.get_trajectories(4, 7, training=False)[0])
    big = allocate_datasource("synthetic_sc2_star_intruders")
    assert (big.conv_input_channels, big.height, big.binary_input_channels) == (4, 128, 4)
    with pytest.raises(ValueError):
        allocate_datasource("minipacman")          # needs the reference's simulator stack
    with pytest.raises(ValueError):
        allocate_datasource("nope")


@pytest.mark.parametrize("name,ctor,state", [
    ("Encoder", lambda m: m.Encoder(16, 3), lambda: WS.encoder_state(0)),
    ("Transition", lambda m: m.Transition(16, 5), lambda: WS.transition_state(0)),
    ("Decoder", lambda m: m.Decoder(16, 3), lambda: WS.decoder_state(0)),
    ("RewardPredictor", lambda m: m.RewardPredictor(16, 2), lambda: WS.reward_state(0)),
    ("Discriminator", lambda m: m.Discriminator(), lambda: WS.discriminator_state(0)),
])
def test_state_dict_layout_matches_reference(name, ctor, state):
    """Key names, shapes and dtypes are the reference's:
    the golden generator loaded exactly these dicts into the reference modules."""
    from scm_gan_amd import models
    net = ctor(models)
    ref = state()
    sd = net.state_dict()
    assert set(sd.keys()) == set(ref.keys()), (sorted(set(sd) ^ set(ref)))
    for k, v in ref.items():
        assert tuple(sd[k].shape) == tuple(np.asarray(v).shape), k
    net.load_state_dict(WS.to_torch(ref))             # strict
    back = net.state_dict()
    for k, v in ref.items():
        assert np.array_equal(back[k].cpu().numpy(), np.asarray(v)), k
    # u, v are parameters that never receive gradients (spectral_normalization.py:53-54)
    for k, p in net.named_parameters():
        if k.endswith("weight_u") or k.endswith("weight_v"):
            assert not p.requires_grad


def test_param_counts():
    from scm_gan_amd import models
    count = lambda m: sum(p.numel() for p in m.parameters())
    assert count(models.Encoder(16, 3)) == 327137 - 1 + 1      # SURVEY 8a (num_batches_tracked is a buffer)
    assert count(models.Transition(16, 5)) == 805581
    assert count(models.Decoder(16, 3)) == 36976
    assert count(models.RewardPredictor(16, 2)) == 6374
    assert count(models.Discriminator()) == 21340


def test_csrn_and_coordconv_state_layout():
    from scm_gan_amd.spatial_recurrent import CSRN
    from scm_gan_amd.coordconv import CoordConv2d
    cs = CSRN(8)
    assert set(cs.state_dict().keys()) == set(WS.csrn_state(0, 8).keys())
    cs.load_state_dict(WS.to_torch(WS.csrn_state(0, 8)))
    cc = CoordConv2d(6 + 2, 10, 3, padding=1)
    assert set(cc.state_dict().keys()) == {"conv.weight", "conv.bias"}
    assert tuple(cc.conv.weight.shape) == (10, 8, 3, 3)


def test_horizon_schedule_and_flags():
    from scm_gan_amd import main as M
    a = M.parser.parse_args([])
    assert (a.batch_size, a.train_iters, a.horizon_min, a.horizon_max, a.learning_rate, a.reward_coef) == \
        (32, 10000, 3, 10, 1e-4, 1e-3)
    # main.py:143-145
    hs = [a.horizon_min + int((a.horizon_max - a.horizon_min) * (it / a.train_iters)) for it in (1, 5000, 10000)]
    assert hs == [3, 6, 10]
    for flag in ("--truncate-bptt", "--latent-overshooting", "--latent-td", "--finetune-reward",
                 "--enable-action-control-loss", "--enable-disentanglement-loss"):
        assert getattr(M.parser.parse_args([flag]), flag[2:].replace("-", "_")) is True


def test_timeseries_standin():
    from scm_gan_amd.logutil import TimeSeries, sparkline
    ts = TimeSeries("t", 10)
    for i in range(3):
        ts.collect("a", torch.tensor(float(i)))
        ts.print_every(10)
    assert ts.values("a") == [0.0, 1.0, 2.0] and "a=2.0000" in str(ts)
    assert len(sparkline([1, 2, 3])) == 3


# ---- the three numpy simulators restated in scm_gan_amd/simulators.py (parity unpinned: rule checks)
def test_gameoflife_step_is_conways_rule_on_a_torus():
    from scipy.signal import convolve2d
    from scm_gan_amd.simulators import GameOfLife
    g = GameOfLife(seed=3)
    st = g.reset(5)
    before = st["cells"].copy()
    r = g.step(st, np.zeros(5, dtype=int))
    kernel = np.ones((3, 3)); kernel[1, 1] = 0
    for b in range(5):
        n = convolve2d(before[b].astype(int), kernel, mode="same", boundary="wrap")      # envs/gameoflife.py:35-36
        assert np.array_equal(st["cells"][b], (n == 3) | (before[b] & (n == 2)))
    assert not r.any()
    s, rw, d, a = g.get_trajectories(3, 6)
    assert s.shape == (3, 6, 1, 64, 64) and rw.shape == (3, 6, 1) and d.shape == (3, 6) and a.shape == (3, 6)
    assert set(np.unique(s)) <= {0.0, 1.0} and not a.any() and not d.any()


def test_gridworld_moves_clamps_and_rewards():
    from scm_gan_amd.simulators import GridWorld
    g = GridWorld(seed=0)
    st = {"x": np.array([8, 14, 2, 8, 8]), "y": np.array([8, 8, 8, 3, 13])}
    r = g.step(st, np.array([0, 3, 2, 0, 1]))      # up, right, left, up, down
    assert st["y"].tolist() == [5, 8, 8, 2, 14]      # y clamps to [2, 14]
    assert st["x"].tolist() == [8, 16, 0, 8, 8]      # x clamps to [0, 16]
    assert r.tolist() == [0, 1, -1, 0, 0]            # right wall +1, left wall -1
    f = g.render({"x": np.array([8, 0, 16]), "y": np.array([8, 8, 8])})
    assert f.shape == (3, 3, 16, 16) and f[:, 0].sum() == 0 and f[:, 2].sum() == 0
    assert f[0, 1, 6:10, 6:10].all() and f[0, 1].sum() == 16          # rows/cols [c-2, c+2)
    assert f[1, 1, 6:10, 0:2].all() and f[1, 1].sum() == 8            # clipped at the left edge
    assert f[2, 1, 6:10, 14:15].all() and f[2, 1].sum() == 4          # right edge: [14, min(18, 15))
    # the action stored at t is the one applied at t+1
    g2 = GridWorld(seed=5)
    s, rw, d, a = g2.get_trajectories(16, 8)
    cy = np.array([[np.argwhere(s[b, t, 1].any(1)).mean() for t in range(8)] for b in range(16)])
    up, down = (a[:, :-1] == 0), (a[:, :-1] == 1)
    dy = cy[:, 1:] - cy[:, :-1]
    assert (dy[up] <= 0).all() and (dy[down] >= 0).all() and (dy[~(up | down)] == 0).all()


def test_pong_bounces_and_scores():
    from scm_gan_amd.simulators import Pong
    g = Pong(seed=0)
    z = lambda *v: np.array(v)
    # ball reaches the right paddle's column in front of the paddle: vx flips; same column away from it: goes on
    st = {"left_y": z(32, 32), "right_y": z(32, 5), "bx": z(54, 54), "by": z(30, 30), "vx": z(2, 2), "vy": z(2, 2)}
    r = g.step(st, z(2, 2))
    assert st["bx"].tolist() == [56, 56] and st["vx"].tolist() == [-2, 2] and not r.any()
    # top / bottom walls
    st = {"left_y": z(32, 32), "right_y": z(32, 32), "bx": z(30, 30), "by": z(61, 4), "vx": z(2, 2), "vy": z(2, -3)}
    g.step(st, z(0, 0))
    assert st["vy"].tolist() == [-2, 3]
    # leaving the court: +1 on the right, -1 on the left, and the ball is sent back
    st = {"left_y": z(5, 5), "right_y": z(5, 5), "bx": z(62, 2), "by": z(40, 40), "vx": z(3, -3), "vy": z(2, 2)}
    r = g.step(st, z(0, 0))
    assert r.tolist() == [1, -1] and st["vx"].tolist() == [-3, 3]
    # buttons: 0/1 red (right) paddle, 2/3 blue (left) paddle, clamped to the court
    st = {"left_y": z(32, 32, 1, 32), "right_y": z(32, 63, 32, 32), "bx": z(30, 30, 30, 30), "by": z(30, 30, 30, 30),
          "vx": z(2, 2, 2, 2), "vy": z(2, 2, 2, 2)}
    g.step(st, z(0, 1, 2, 3))
    assert st["right_y"].tolist() == [29, 64, 32, 32] and st["left_y"].tolist() == [32, 32, 0, 35]
    f = g.render({"left_y": z(32), "right_y": z(20), "bx": z(40), "by": z(10), "vx": z(2), "vy": z(2)})
    assert f[0, 2, 24:40, 4:6].all() and f[0, 2].sum() == 32          # blue paddle, left
    assert f[0, 0, 12:28, 58:60].all() and f[0, 0].sum() == 32        # red paddle, right
    assert f[0, 1, 8:12, 38:42].all() and f[0, 1].sum() == 16         # green ball
    s, rw, d, a = g.get_trajectories(4, 5)
    assert s.shape == (4, 5, 3, 64, 64) and rw.shape == (4, 5, 1) and a.max() <= 3


def test_allocate_datasource_numpy_games():
    from scm_gan_amd.datasource import allocate_datasource
    for name, (c, size, acts) in {"pong": (3, 64, 4), "gridworld": (3, 16, 4), "gameoflife": (1, 64, 1)}.items():
        ds = allocate_datasource(name, seed=2)
        assert (ds.conv_input_channels, ds.height, ds.binary_input_channels, ds.scalar_output_channels) == (c, size, acts, 1)
        a1 = allocate_datasource(name, seed=2).get_trajectories(2, 3)
        a2 = ds.get_trajectories(2, 3)
        assert all(np.array_equal(x, y) for x, y in zip(a1, a2))          # seeded: reproducible
