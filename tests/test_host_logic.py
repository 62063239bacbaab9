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
