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
    enc.load_state_dict(WS.to_torch(WS.encoder_state(seed, color=C)))
    tr.load_state_dict(WS.to_torch(WS.transition_state(seed + 1, actions=A)))
    dec.load_state_dict(WS.to_torch(WS.decoder_state(seed + 2, color=C)))
    rp.load_state_dict(WS.to_torch(WS.reward_state(seed + 3, rewards=R)))
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


def _loss_and_grads(nets, states, rewards, dones, onehots, noise, theta=0.5, coef=1e-3):
    """One unrolled training loss (main.py:162-215) through the HIP modules; returns flat grads."""
    from scm_gan_amd import ops
    enc, dec, rp, tr = nets
    for m in nets:
        m.train()
        for p in m.parameters():
            p.grad = None
    tr.set_noise([n.clone() for n in noise])
    z = enc(states[:, 0:3])
    mask = torch.ones(states.shape[0], device="cuda")
    loss = None
    for t in range(1, states.shape[1] - 1):
        r = rp(z)
        bce = ops.pixel_bce(dec(z), states[:, t].contiguous())
        loss, _, _, mask = ops.StepLossFn.apply(loss, r, bce, rewards[:, t], dones[:, t], mask, theta * coef)
        z = tr(z, onehots[:, t])
    loss.backward()
    return torch.cat([p.grad.flatten() for m in nets for p in m.parameters() if p.grad is not None]), float(loss.detach())


def test_data_parallel_shards_average_to_full_batch():
    """equal shards of the batch, gradients summed and scaled by 1/world == the
    full-batch gradient (what the RCCL all-reduce + AdamClip.grad_scale implement), checked on
    the HIP operators themselves with two 'virtual ranks' on one GPU."""
    from scm_gan_amd import main as T
    from scm_gan_amd.datasource import SyntheticDatasource
    enc, dec, rp, disc, tr = build(70)
    nets = (enc, dec, rp, tr)
    B, h, H, W, A = 8, 5, 16, 16, 5
    ds = SyntheticDatasource(seed=9, height=H, width=W, done_every=3)
    s, r, d, a = ds.get_trajectories(B, h)
    states, rewards = dev(s.astype(np.float32)), dev(r.astype(np.float32))
    dones, onehots = dev(d.astype(np.float32)), T.onehot(a, A)
    rng = np.random.default_rng(1)
    noise = [dev(rng.random((B, 16, H, W), dtype=np.float32)) for _ in range(h - 2)]
    sd = [{k: v.clone() for k, v in m.state_dict().items()} for m in nets]

    def reset():
        for m, st in zip(nets, sd):
            m.load_state_dict(st)          # SpectralNorm u/v restart from the same state for every run
    reset()
    g_full, l_full = _loss_and_grads(nets, states, rewards, dones, onehots, noise)
    parts, losses = [], []
    for k in range(2):
        reset()
        sl = slice(4 * k, 4 * k + 4)
        g, l = _loss_and_grads(nets, states[sl].contiguous(), rewards[sl].contiguous(), dones[sl].contiguous(),
                               onehots[sl].contiguous(), [n[sl].contiguous() for n in noise])
        parts.append(g)
        losses.append(l)
    g_dp = (parts[0] + parts[1]) / 2
    assert abs((losses[0] + losses[1]) / 2 - l_full) < 1e-5 * max(1.0, abs(l_full))
    assert float((g_dp - g_full).abs().max()) < 2e-5 * max(1.0, float(g_full.abs().max()))


def test_short_training_run_is_stable_and_learns():
    """40 iterations of train() on a fixed tiny batch: finite, and the reconstruction loss drops."""
    from scm_gan_amd import main as T
    from scm_gan_amd.datasource import SyntheticDatasource

    class Fixed(SyntheticDatasource):
        def get_trajectories(self, *a, **k):
            if not hasattr(self, "_b"):
                self._b = SyntheticDatasource.get_trajectories(self, *a, **k)
            return self._b
    enc, dec, rp, disc, tr = build(80)
    tr.seed_noise(5)
    ds = Fixed(seed=2, height=16, width=16)
    T.args = T.parser.parse_args(["--batch-size", "8", "--train-iters", "40", "--horizon-min", "5", "--horizon-max", "5",
                                  "--eval-every", "1000000", "--learning-rate", "0.001", "--log-every", "1000"])
    ts = T.train(16, ds, 5, 2, enc, dec, rp, disc, tr)
    T.args = T.parser.parse_args([])
    rec = np.array(ts.values("Reconstruction t=1"))
    assert np.all(np.isfinite(rec))
    assert rec[-5:].mean() < 0.8 * rec[:5].mean(), (rec[:5].mean(), rec[-5:].mean())
    for m in (enc, dec, rp, tr):
        for p in m.parameters():
            assert torch.isfinite(p).all()


def test_excitation_backtrack_runs_and_attributes():
    """excitation_bptt.py:6-53 numeric core: returns per-step action attributions and latent masks."""
    from scm_gan_amd.excitation_bptt import excitation_backtrack
    enc, dec, rp, disc, tr = build(90, A=4)
    tr.train()
    tr.seed_noise(3)
    z = (torch.rand(1, 16, 16, 16, device="cuda") < 0.5).float()
    out = excitation_backtrack(z, tr, rp, 4, steps=6, threshold=0.0)
    assert out is not None and out["t"] == 0
    assert len(out["action_attributions"]) == 1 and out["action_attributions"][0].shape == (1, 4)
    m = out["latent_masks"][0]
    assert m.shape == z.shape and float(m.max()) <= 1.0 and torch.isfinite(m).all()


def test_graphed_planner_matches_eager_and_is_faster():
    """planner.GraphedRollout: the HIP-graph replay of compute_rollout_reward equals the eager
    rollout started from the same SpectralNorm state, call after call."""
    import copy
    import time
    from scm_gan_amd import main as T
    from scm_gan_amd.planner import GraphedRollout
    enc, dec, rp, disc, tr = build(110)
    T.test_mode([tr, rp])
    tr2, rp2 = copy.deepcopy(tr), copy.deepcopy(rp)
    rng = np.random.default_rng(4)
    zs = [dev((rng.random((1, 16, 32, 32)) < 0.5).astype(np.float32)) for _ in range(3)]
    planner = GraphedRollout(tr, rp, 5)
    # the graph's two warm-up rollouts + capture advance u/v: replay the same number of eager rollouts on the twin
    with torch.no_grad():
        got0 = planner(zs[0])
        for _ in range(3):
            T.compute_rollout_reward(zs[0], tr2, rp2, 5, selected_action=0)
        for z in zs:
            want = T.compute_rollout_reward(z, tr2, rp2, 5, selected_action=0)
            got = planner(z)
            assert abs(float(got) - float(want)) < 2e-3 * max(1.0, abs(float(want))), (float(got), float(want))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            planner(zs[0])
        torch.cuda.synchronize()
        t_graph = (time.perf_counter() - t0) / 5
        t0 = time.perf_counter()
        for _ in range(5):
            T.compute_rollout_reward(zs[0], tr2, rp2, 5, selected_action=0)
        torch.cuda.synchronize()
        t_eager = (time.perf_counter() - t0) / 5
    print("planner: graph %.2f ms  eager %.2f ms" % (t_graph * 1e3, t_eager * 1e3))
    assert t_graph < t_eager


def test_graphed_planner_recaptures_after_a_weight_update():
    import copy
    from scm_gan_amd import main as T
    from scm_gan_amd.planner import GraphedRollout
    enc, dec, rp, disc, tr = build(111)
    T.test_mode([tr, rp])
    z = dev((np.random.default_rng(6).random((1, 16, 32, 32)) < 0.5).astype(np.float32))
    planner = GraphedRollout(tr, rp, 5)
    with torch.no_grad():
        r0 = float(planner(z))
        rp.conv2.bias.add_(0.25)                       # an in-place update bumps the parameter version
        tr2, rp2 = copy.deepcopy(tr), copy.deepcopy(rp)
        got = float(planner(z))                        # recapture: 2 warm-up rollouts + capture + replay = 4 rollouts
        for _ in range(3):
            T.compute_rollout_reward(z, tr2, rp2, 5, selected_action=0)
        want = float(T.compute_rollout_reward(z, tr2, rp2, 5, selected_action=0))
    assert abs(got - want) < 2e-3 * max(1.0, abs(want)), (got, want, r0)
    assert abs(got - r0) > 1e-3


def test_device_batch_queue_matches_the_serial_loop(tmp_path, monkeypatch):
    """train() fed by datasource.DeviceBatchQueue (worker thread, pinned double-buffered uploads, on-device
    one-hot; --prefetch 2) ends with BITWISE the weights of the serial loop, growing horizon included."""
    from scm_gan_amd import main as T, models as M, _lib as L
    from scm_gan_amd.datasource import SyntheticDatasource
    import ctypes
    monkeypatch.chdir(tmp_path)
    # the one-hot kernel against the reference formulation (eye[idx])
    idx = torch.randint(0, 5, (7, 11), device="cuda")
    oh = torch.empty(7, 11, 5, device="cuda")
    L.check(L.lib().scm_onehot(ctypes.c_void_p(idx.data_ptr()), idx.numel(), 5, ctypes.c_void_p(oh.data_ptr()),
                               ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "scm_onehot")
    assert torch.equal(oh, torch.eye(5, device="cuda")[idx])

    def run(prefetch):
        torch.manual_seed(21)
        enc, dec, rp, disc, tr = M.Encoder(16, 3), M.Decoder(16, 3), M.RewardPredictor(16, 2), M.Discriminator(), \
            M.Transition(16, 5)
        tr.seed_noise(9)
        ds = SyntheticDatasource(seed=4, height=32, width=32, channels=3, num_actions=5, num_rewards=2, done_every=3)
        T.args = T.parser.parse_args(["--batch-size", "6", "--train-iters", "7", "--horizon-min", "3", "--horizon-max", "8",
                                      "--eval-every", "1000000", "--prefetch", str(prefetch)])
        T.train(16, ds, 5, 2, enc, dec, rp, disc, tr)
        return torch.cat([p.detach().flatten() for m in (enc, tr, dec, rp) for p in m.parameters()])

    w_serial = run(0)
    w_queue = run(2)
    assert torch.equal(w_serial, w_queue)


def test_play_agent_on_gridworld_matches_the_eager_planner(tmp_path, monkeypatch):
    """main.play() (reference main.py:327-400) on the restated gridworld game: the HIP-graph planner and the
    eager compute_rollout_reward pick the same actions from the same weights and game seed."""
    import copy
    from scm_gan_amd import main as T, models as M
    from scm_gan_amd.datasource import allocate_datasource
    monkeypatch.chdir(tmp_path)
    torch.manual_seed(8)
    nets = [M.Encoder(16, 3), M.Decoder(16, 3), M.RewardPredictor(16, 1), M.Discriminator(), M.Transition(16, 4)]
    runs = []
    for use_graph in (True, False):
        enc, dec, rp, disc, tr = copy.deepcopy(nets)
        ds = allocate_datasource("gridworld", seed=3)
        t, total, chosen = T.play(16, ds, 4, 1, enc, dec, rp, disc, tr, max_steps=12, use_graph=use_graph, verbose=False)
        runs.append((t, total, chosen))
        assert t == 13 and len(chosen) == 11 and all(0 <= a < 4 for a in chosen)
    # SpectralNorm's power iteration advances with every forward call and the graph adds warm-up rollouts, so the two
    # agents see slightly different sigma: the first decisions (before that drift matters) must agree
    assert runs[0][2][:3] == runs[1][2][:3], runs
    assert list(tmp_path.glob("evaluation_metrics_*.txt"))


def test_compute_rollout_reward_vs_reference_run():
    """main.compute_rollout_reward against the reference's OWN main.compute_rollout_reward run (tests/golden/planner.npz: A=5 at 16x16, A=4 at 12x12): every reward-head
    output along the 12-step beam (forward hook) and the best cumulative reward."""
    from scm_gan_amd import main as T, models as M
    fx = load("planner.npz")
    for A in (5, 4):
        tr, rp = M.Transition(16, A), M.RewardPredictor(16, 2)
        tr.load_state_dict(WS.to_torch(WS.transition_state(200 + A, actions=A)))
        rp.load_state_dict(WS.to_torch(WS.reward_state(210 + A, rewards=2)))
        T.test_mode([tr, rp])
        seen = []
        hook = rp.register_forward_hook(lambda m, i, o: seen.append(o.detach().cpu().numpy().copy()))
        with torch.no_grad():
            best = T.compute_rollout_reward(dev(fx["A%d.z" % A]), tr, rp, A, 0, rollout_depth=12, rollout_policy='noop')
        hook.remove()
        got = np.stack(seen)
        ref = fx["A%d.rewards" % A]
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() < 1e-4 * max(1.0, np.abs(ref).max()), np.abs(got - ref).max()
        assert abs(float(best) - float(fx["A%d.best" % A])) < 5e-4 * max(1.0, abs(float(fx["A%d.best" % A])))


def test_excitation_backtrack_vs_reference_run():
    """excitation_bptt.excitation_backtrack against the reference's OWN excitation_bptt.visualize_bptt run
    (tests/golden/bptt.npz): the step the 0.8 threshold is crossed at, rewards, action attributions
    a_t * dr/da_t and the latent masks |dz_t| / (.001 + max|dz_t|)."""
    from scm_gan_amd import models as M
    from scm_gan_amd.excitation_bptt import excitation_backtrack
    fx = load("bptt.npz")
    tr, rp = M.Transition(16, 4), M.RewardPredictor(16, 2)
    tr.load_state_dict(WS.to_torch(WS.transition_state(300, actions=4)))
    rp.load_state_dict(WS.to_torch(WS.reward_state(301, rewards=2, gain=0.31)))
    tr.train()
    tr.set_noise([dev(n) for n in fx["noise"]])
    out = excitation_backtrack(dev(fx["z0"]), tr, rp, 4, steps=30, threshold=0.8)
    steps = int(fx["steps"])
    assert out is not None and out["t"] == steps - 1
    assert np.abs(out["reward"].cpu().numpy() - fx["rewards"][-1]).max() < 1e-4
    for t in range(steps):
        want = fx["a"][t] * fx["a_grad"][t]
        got = out["action_attributions"][t].cpu().numpy()
        assert np.abs(got - want).max() < 1e-6 + 2e-3 * np.abs(fx["a_grad"]).max(), (t, got, want)
    for i, t in enumerate(range(steps - 1, -1, -1)):            # latest step first
        g = np.abs(fx["z_grad"][t])
        want = g / (.001 + g.max())
        got = out["latent_masks"][i].cpu().numpy()
        assert np.abs(got - want).max() < 2e-3, (t, np.abs(got - want).max())
