"""Train / rollout API of the hot path, with the reference's function signatures
(/root/reference/main.py): train(), evaluate(), measure_prediction_mse(), play(),
compute_rollout_reward(), onehot(), train_mode(), test_mode().

The per-iteration body follows main.py:143-297 line by line; the networks are
scm_gan_amd.models (HIP operators), the optimizer step is the fused
clip+Adam kernel, and under torch.distributed (one process per GPU, RCCL) the
gradients are summed with one all-reduce per network over a flat bucket before
the clip, so N ranks with batch B/N each reproduce one rank with batch B.

Out of scope here: the mp4/png visualisers (play() keeps its
numeric agent loop and drops the video writer), causal-graph rendering.
"""
import argparse
import json
import os
import time

import numpy as np
import torch
import torch.nn as nn

from . import dp
from . import models
from . import ops
from .datasource import allocate_datasource
from .logutil import TimeSeries, sparkline
from .optim import AdamClip
from . import spectral_normalization as SN

parser = argparse.ArgumentParser(description="Learn to model a sequential environment")
parser.add_argument('--env', default='synthetic_minipacman', help='Datasource name (see scm_gan_amd/datasource.py)')
parser.add_argument('--load-from', type=str, help='Directory containing .pth models to load before starting')
parser.add_argument('--evaluate', action='store_true', help='If true, evaluate instead of training')
parser.add_argument('--evaluations', type=int, default=1, help='Integer number of evaluations to run')
parser.add_argument('--title', type=str, help='Name of experiment in output figures')
parser.add_argument('--batch-size', type=int, default=32, help='Training batch size')
parser.add_argument('--train-iters', type=int, default=10000, help='Number of iterations of training')
parser.add_argument('--start-iter', type=int, default=1, help='Start iteration when resuming from checkpoint')
parser.add_argument('--truncate-bptt', action='store_true', help='Train only with timestep-local information')
parser.add_argument('--latent-overshooting', action='store_true', help='Latent Overshooting (Hafner et al.)')
parser.add_argument('--latent-td', action='store_true', help='(dead flag in the reference, kept for CLI parity)')
parser.add_argument('--td-lambda', type=float, default=0.9, help='Scalar lambda for overshooting')
parser.add_argument('--td-steps', type=int, default=3, help='(dead flag in the reference)')
parser.add_argument('--horizon-min', type=int, default=3, help='Min timestep horizon value')
parser.add_argument('--horizon-max', type=int, default=10, help='Max timestep horizon value')
parser.add_argument('--learning-rate', type=float, default=.0001, help='Adam lr value')
parser.add_argument('--finetune-reward', action='store_true', help='Train ONLY the reward estimation network')
parser.add_argument('--reward-coef', type=float, default=.001, help='Reward loss magnitude')
parser.add_argument('--activation-l1-coef', type=float, default=.01, help='(dead flag in the reference)')
parser.add_argument('--transition-l1-coef', type=float, default=.01, help='(dead flag in the reference)')
parser.add_argument('--enable-action-control-loss', action='store_true', help='CF Action Control regularization')
parser.add_argument('--enable-disentanglement-loss', action='store_true', help='CF Disentanglement regularization')
parser.add_argument('--counterfactual-horizon', type=int, default=1, help='Forward horizon for CF generation')
# additions of this build
parser.add_argument('--seed', type=int, default=0, help='Seed of weights, data and the Philox latent noise')
parser.add_argument('--precision', choices=('fp32', 'fp32x3', 'fp16'), default='fp32',
                    help="MFMA operand precision of the 3x3 convs: fp32 = the reference's arithmetic; fp16 = fp16 forward / "
                         "bf16 gradient operands with fp32 accumulation (not in the reference)")
parser.add_argument('--conv-algo', choices=('winograd', 'direct'), default=None,
                    help="fp32 algorithm of the wide 3x3 convs (both fp32 operands / fp32 accumulation): winograd = F(2x2,3x3), "
                         "4/9 of the MFMA products (default); direct = implicit GEMM")
parser.add_argument('--prefetch', type=int, default=0,
                    help='Depth of the device-side batch queue (datasource.DeviceBatchQueue: worker thread, pinned '
                         'double-buffered uploads, on-device one-hot); 0 = the reference\'s serial loop')
parser.add_argument('--eval-every', type=int, default=2000, help='Evaluate + checkpoint cadence (ITERS_PER_VIDEO)')
parser.add_argument('--log-every', type=int, default=10)
parser.add_argument('--gan', action='store_true',
                    help='BASELINE configs[2]: also run the spectral-normalised discriminator forward / backward / Adam every '
                         'iteration on (B, 3, 64, 64) U{0,1} noise maps (SURVEY.md 8(d) cfg3). The reference constructs the '
                         'discriminator and its optimizer (main.py:76,128) but defines NO GAN loss; the critic term here '
                         '(mean logit) is build-defined')
parser.add_argument('--graph-step', action='store_true',
                    help='replay the training iteration as one HIP graph per (batch shape, horizon) after two eager '
                         'iterations of that shape (graph_step.GraphedTrainer; default loss, single process only)')
parser.add_argument('--no-overlap', action='store_true',
                    help='data parallel: all-reduce every bucket after backward() instead of overlapping with it')
parser.add_argument('--check-replicas', action='store_true',
                    help='data parallel: verify after every evaluation that all ranks hold identical weights and u / v')

# The reference parses sys.argv at import (main.py:51); here the module-level
# `args` holds the defaults and main() replaces it, so the API is importable.
args = parser.parse_args([])

ITERS_PER_VIDEO = 2000
CF_REGULARIZATION_RATE = 5
CF_REGULARIZATION_LAMBDA = .01


def _device():
    return torch.device('cuda', torch.cuda.current_device())


def _upload(a, dtype=torch.float32):
    """numpy (any float dtype) -> fp32 device tensor (main.py:156-158)."""
    t = torch.as_tensor(np.ascontiguousarray(a)).to(dtype)
    return t.to(_device(), non_blocking=True)


def onehot(a_idx, num_actions=4):
    """main.py:447-452"""
    eye = torch.eye(num_actions, device=_device())
    if type(a_idx) is int:
        return eye[a_idx].unsqueeze(0)
    idx = torch.as_tensor(np.asarray(a_idx) if not torch.is_tensor(a_idx) else a_idx).long().to(_device())
    return eye[idx]


def test_mode(networks):
    """main.py:492-497"""
    for net in networks:
        net.eval()
        for child in net.children():
            if type(child) == nn.BatchNorm2d or type(child) == nn.BatchNorm1d:
                child.momentum = 0


def train_mode(networks):
    """main.py:500-505"""
    for net in networks:
        net.train()
        for child in net.children():
            if type(child) == nn.BatchNorm2d or type(child) == nn.BatchNorm1d:
                child.momentum = 0.1


def latent_state_loss(target, predicted):
    """main.py:306-307"""
    return ((target - predicted) ** 2).mean(-1).mean(-1).mean(-1)


def decoder_pixel_loss(target, predicted_logits):
    """main.py:310-312 fused with the torch.sigmoid of main.py:189: takes the
    decoder LOGITS and returns the per-sample mean binary cross entropy."""
    return ops.pixel_bce(predicted_logits, target)


_FlatGrads = dp.FlatGrads        # (kept under the round-1 names: tests import them)
_dist = dp.world_info


def decoder_frame_loss(decoder, z, target, kind=0):
    """Per-sample loss of the decoded frame: main.py:188-190 (kind 0: sigmoid + decoder_pixel_loss) or the squared
    error of main.py:814-816 (kind 1).  A decoder with forward_loss (scm_gan_amd.models.Decoder) computes it in one
    fused pass over z; any other module goes through its forward."""
    if hasattr(decoder, 'forward_loss'):
        return decoder.forward_loss(z, target, kind)
    logits = decoder(z)
    return decoder_pixel_loss(target, logits) if kind == 0 else ops.pixel_mse(logits, target)


def _heads_time_batched(a, decoder, reward_predictor, z, nsteps):
    """Whether Trainer.step runs the reward head and the decoder once over all unrolled steps (time-batched) instead
    of once per step: for this package's own modules (fused kernels or layer by layer: both measured faster -- 64x64
    +3.9 %, 84x84 +1.3 %, 128x128x4 +1.7 % / +3.0 % in the 16-bit mode), when more than one step is unrolled, no option
    reads the running mask inside the loop (--latent-overshooting does) and the widest time-batched map (64 channels)
    stays below 2^31 elements.  SCM_BATCHED_HEADS=0 keeps the per-step launches (A/B), =1 batches whatever the modules
    are (CPU host-logic tests)."""
    mode = os.environ.get("SCM_BATCHED_HEADS", "")
    if mode == "0" or nsteps < 2 or a.latent_overshooting:
        return False
    if mode == "1":
        return True
    if not (hasattr(decoder, "_fused") and hasattr(reward_predictor, "_fused")):
        return False
    return nsteps * z.shape[0] * 64 * z.shape[2] * z.shape[3] < 2 ** 31


class Trainer(object):
    """One training iteration of main.py:143-297 on device-resident batches.

    train() below feeds it from a datasource exactly as the reference does;
    bench.py feeds it batches that already live in HBM."""

    def __init__(self, latent_dim, num_actions, num_rewards, encoder, decoder, reward_predictor, discriminator,
                 transition, opts=None):
        self.a = opts if opts is not None else args
        self.latent_dim, self.num_actions, self.num_rewards = latent_dim, num_actions, num_rewards
        self.encoder, self.decoder, self.reward_predictor = encoder, decoder, reward_predictor
        self.discriminator, self.transition = discriminator, transition
        self.rank, self.world = _dist()
        lr = self.a.learning_rate
        # main.py:125-129 + 287-290: clip 0.1 on enc/trans/dec, none on the reward head;
        # the discriminator's optimizer is created but never stepped in the reference.
        self.opt_enc = AdamClip(encoder.parameters(), lr=lr, clip_value=0.1)
        self.opt_dec = AdamClip(decoder.parameters(), lr=lr, clip_value=0.1)
        self.opt_trans = AdamClip(transition.parameters(), lr=lr, clip_value=0.1)
        self.opt_disc = AdamClip(discriminator.parameters(), lr=lr)
        self.opt_pred = AdamClip(reward_predictor.parameters(), lr=lr)
        self.gan = bool(getattr(self.a, 'gan', False))
        self.flat_enc = dp.FlatGrads(encoder, 'encoder')
        # complete once the gradient of the first encoder output exists (dp.GradSync)
        self.flats_early = [dp.FlatGrads(decoder, 'decoder'), dp.FlatGrads(transition, 'transition'),
                            dp.FlatGrads(reward_predictor, 'reward_predictor')]
        if self.gan:
            self.flats_early.append(dp.FlatGrads(discriminator, 'discriminator'))
        self.flats = [self.flat_enc] + self.flats_early
        for o in (self.opt_enc, self.opt_dec, self.opt_trans, self.opt_pred, self.opt_disc):
            o.grad_scale = 1.0 / self.world
        force = os.environ.get("SCM_FORCE_ALLREDUCE") == "1"
        overlap = not getattr(self.a, 'no_overlap', False)
        self.sync = dp.GradSync(self.flats_early if overlap else [], [self.flat_enc] if overlap else self.flats,
                                enabled=self.world > 1 or force)
        self.ts = TimeSeries('Training Model', self.a.train_iters, tensorboard=True)
        self.side_stream = torch.cuda.Stream()
        self._disc_half = None
        self._disc_noise_offset = 0

    def step(self, states, rewards, dones, onehots, theta, train_iter=1, actions=None, disc_noise=None):
        """states (B,T,C,H,W), rewards (B,T,R), dones (B,T) as 0/1 floats, onehots
        (B,T,A): fp32 device tensors; disc_noise: optional (B,3,64,64) U{0,1} maps for the
        --gan discriminator pass (drawn on the device otherwise).  Returns the (device) loss."""
        a = self.a
        ts = self.ts
        encoder, decoder, reward_predictor, transition = self.encoder, self.decoder, self.reward_predictor, \
            self.transition
        batch_size, prediction_horizon = states.shape[0], states.shape[1]
        latent_dim = self.latent_dim
        cf_now = train_iter % CF_REGULARIZATION_RATE == 0
        enable_cf_shuffle_loss = a.enable_disentanglement_loss and cf_now
        enable_control_bias_loss = a.enable_action_control_loss and cf_now

        train_mode([encoder, decoder, transition, self.discriminator, reward_predictor])
        for f in self.flats:
            f.zero()

        # SpectralNorm power iterations depend on the weights only: the encoder's three
        # (and the discriminator's two) run in one launch, and the (horizon-2) x 5 of the
        # unrolled transition run on a side stream underneath the encoder's convolutions.
        # Sigmas an aborted iteration left queued are dropped first.
        SN.clear_precomputed([encoder, transition, self.discriminator])
        main_stream = torch.cuda.current_stream()
        self.side_stream.wait_stream(main_stream)           # (the previous iteration's Adam)
        # (two launches: the first transition call's iterations -- ~0.2 ms on 5 CUs, ready well before the encoder ends --
        # and the remaining calls', which then have until the SECOND transition call; as one launch of 1.36 ms next to
        # the 1.1 ms encoder the main stream waited 0.17 ms for it every iteration)
        ncalls = prediction_horizon - 2
        with torch.cuda.stream(self.side_stream):
            first = ncalls if (ncalls <= 1 or os.environ.get("SCM_SN_ONE_LAUNCH", "0") not in ("", "0")) else 1
            SN.precompute([transition], first)
            self.sn_first = torch.cuda.Event()
            self.sn_first.record(self.side_stream)
            if ncalls > first:
                SN.precompute([transition], ncalls - first)
        SN.precompute([encoder] + ([self.discriminator] if self.gan else []), 1)

        z = encoder(states[:, 0:3])                                            # main.py:162
        self.sync.arm(z)
        main_stream.wait_event(self.sn_first)
        targets = states.transpose(0, 1).contiguous()      # (T, B, C, H, W): targets[t] is one contiguous frame batch
        onehots_t = onehots.transpose(0, 1).contiguous()   # (T, B, A): one copy instead of one per unrolled step
        z_orig = z.clone() if (enable_cf_shuffle_loss or enable_control_bias_loss) else None
        active_mask = torch.ones(batch_size, device=states.device)

        loss = 0
        lo_loss = 0
        lo_z_set = {}
        # Time-batched heads: the reward head and the decoder + pixel loss of main.py:181-190 consume z_t and feed
        # nothing back into the rollout, so the steps' latents are collected and both run ONCE over (horizon-2) x B
        # samples behind the loop -- same per-sample arithmetic, same per-step losses in the same order; what changes
        # is that one launch covers whole images per workgroup instead of 8-row bands with their halo rows
        # (tools/bench_heads.py: 8 x 64 samples 4.8 ms, 1 x 512 samples 3.5 ms forward + backward).
        heads_batched = _heads_time_batched(a, decoder, reward_predictor, z, prediction_horizon - 2)
        zs = []
        for t in range(1, prediction_horizon - 1):                             # main.py:177
            if heads_batched:
                zs.append(z)
            else:
                expected_reward = reward_predictor(z)                              # main.py:181
                rec_loss_batch = decoder_frame_loss(decoder, z, targets[t])        # main.py:188-190
            if a.truncate_bptt and t > 1:
                z = z.detach()
            if not heads_batched:
                # main.py:178,182-185,195-197 in one launch: mask update, both masked means, loss accumulation
                loss, reward_difference, rec_loss, active_mask = ops.StepLossFn.apply(
                    loss if torch.is_tensor(loss) else None, expected_reward, rec_loss_batch, rewards[:, t], dones[:, t],
                    active_mask, theta * a.reward_coef)
                ts.collect('Rd Loss t={}'.format(t), reward_difference)
                ts.collect('Reconstruction t={}'.format(t), rec_loss)

            if t == 2:
                main_stream.wait_stream(self.side_stream)                      # (the remaining calls' power iterations)
            z = transition(z, onehots_t[t])                                    # main.py:206-215

            if a.latent_overshooting:
                # main.py:217-230 (structure kept, incl. its index quirks)
                lo_z_set[t] = encoder(states[:, t - 1:t + 2])
                for t_left in range(1, t):
                    lo_z_set[t_left] = transition(lo_z_set[t_left], onehots[:, t - 1])
                for t_a in range(2, t - 1):
                    lo_loss_batch = latent_state_loss(lo_z_set[t].detach(), lo_z_set[t_a])
                    lo_loss += a.td_lambda * torch.mean(lo_loss_batch * active_mask)

        if heads_batched:
            nT = len(zs)
            z_all = torch.cat(zs, 0)                                           # (nT * B, L, H, W), step-major
            frames_all = targets[1:1 + nT].reshape((nT * batch_size,) + tuple(targets.shape[2:]))
            expected_reward = reward_predictor(z_all)                          # main.py:181, all steps
            rec_loss_batch = decoder_frame_loss(decoder, z_all, frames_all)    # main.py:188-190, all steps
            loss, rd_rec, active_mask = ops.StepLossSeqFn.apply(expected_reward, rec_loss_batch, rewards, dones, 1, nT,
                                                                active_mask, theta * a.reward_coef)
            for i in range(nT):
                ts.collect('Rd Loss t={}'.format(1 + i), rd_rec[i, 0:1])
                ts.collect('Reconstruction t={}'.format(1 + i), rd_rec[i, 1:2])
            del zs, z_all

        if a.latent_overshooting:
            ts.collect('LO total', lo_loss)
            loss = loss + theta * lo_loss

        if enable_cf_shuffle_loss:
            cf_loss = self._cf_disentanglement(z, z_orig, onehots, active_mask)
            loss = loss + cf_loss
            ts.collect('CF Disentanglement Loss', cf_loss)

        if enable_control_bias_loss:
            cf_loss = self._cf_action_control(z, z_orig, actions, active_mask)
            loss = loss + cf_loss
            ts.collect('CF Control Bias Loss', cf_loss)

        if self.gan:
            d_loss = self._discriminator_loss(batch_size, disc_noise)
            ts.collect('D mean logit', d_loss)

        with ops.grad_sinks():             # weight / bias gradients accumulate straight into the flat buckets
            torch.autograd.backward([loss, d_loss] if self.gan else [loss])    # main.py:285

        # data parallel: SUM over ranks (the early buckets were issued from the hook on z, under the encoder's
        # backward); AdamClip applies 1/world before the clip, so N ranks x B/N equal one rank x B
        self.sync.finish()

        self.opt_pred.step()                                                   # main.py:292-296
        if not a.finetune_reward:
            self.opt_enc.step()
            self.opt_dec.step()
            self.opt_trans.step()
        if self.gan:
            self.opt_disc.step()
        return loss.detach()

    def _discriminator_loss(self, batch_size, noise=None):
        """BASELINE configs[2] / SURVEY.md 8(d) cfg3: the discriminator (models.py:195-223) forward on
        (B, NOISE_DIM, 64, 64) maps of U{0,1} noise; its backward and Adam step follow with the rest of
        the iteration.  The reference never forwards its discriminator and has no GAN loss
        (main.py:76,128): the scalar trained on, the mean logit, is build-defined."""
        if noise is None:
            shape = (batch_size, models.NOISE_DIM, 64, 64)     # the only size fc1 accepts: 64 -> 31 -> 15 -> 7
            if self._disc_half is None or self._disc_half.shape != shape:
                self._disc_half = torch.full(shape, 0.5, device=_device())
            seed = getattr(self.transition, 'noise_seed', 0) + 7919          # rank-offset like the latent noise
            noise = ops.bernoulli_sample(self._disc_half, seed, self._disc_noise_offset)
            self._disc_noise_offset += (self._disc_half.numel() + 3) // 4
        return ops.mean_all(self.discriminator(noise))


    def _rollout_counterfactual(self, z_start, step_onehots):
        """Roll the encoded start state forward under a (counterfactual) action plan for
        --counterfactual-horizon - 1 steps (main.py:255-257, 276-278)."""
        z_cf = z_start
        for t in range(1, self.a.counterfactual_horizon):
            z_cf = self.transition(z_cf, step_onehots[:, t])
        return z_cf

    def _cf_disentanglement(self, z_factual, z_start, onehots, active_mask):
        """Counterfactual disentanglement regulariser, main.py:242-262: swap two random latent
        factors per sample in the start state (IN PLACE on z_start, which the reference also
        mutates through its alias), replay the same actions, and penalise any change of the
        factors that were not swapped."""
        n, factors = z_factual.shape[0], self.latent_dim
        keep = torch.ones((n, factors), device=z_factual.device)
        for row in range(n):
            fa, fb = np.random.randint(factors), np.random.randint(factors)
            keep[row, fa] = 0
            keep[row, fb] = 0
            z_start[row, fa], z_start[row, fb] = z_start[row, fb], z_start[row, fa]
        z_swapped = self._rollout_counterfactual(z_start, onehots)
        drift = torch.abs(z_factual.clone() - z_swapped).mean(-1).mean(-1) * keep
        return CF_REGULARIZATION_LAMBDA * torch.mean(drift.mean(-1) * active_mask)

    def _cf_action_control(self, z_factual, z_start, actions, active_mask):
        """Counterfactual action-control regulariser, main.py:268-283: replay the start state
        under a shuffled action plan; the outcome should differ (negative log distance)."""
        plan = np.array(actions).copy()
        np.random.shuffle(plan)
        z_other = self._rollout_counterfactual(z_start, onehot(plan, self.num_actions))
        gap = torch.abs(z_factual.clone() - z_other).mean(-1).mean(-1).mean(-1)
        return CF_REGULARIZATION_LAMBDA * torch.mean(-torch.log(gap + .001) * active_mask)


def train(latent_dim, datasource, num_actions, num_rewards,
          encoder, decoder, reward_predictor, discriminator, transition, eval_datasource=None):
    """main.py:106-299.  Under torch.distributed (one process per GPU) --batch-size is the GLOBAL
    batch: every rank draws batch_size / world trajectories from its own datasource stream.
    eval_datasource: what the periodic evaluation draws from (default: `datasource`).  Data-parallel runs pass
    one that is seeded IDENTICALLY on every rank: the evaluation rollout stops early when every trajectory is
    done (main.py:805), so ranks that evaluated different data could make a different number of SpectralNorm
    forwards and their u / v -- then their weights -- would drift apart."""
    train_iters = args.train_iters
    trainer = Trainer(latent_dim, num_actions, num_rewards, encoder, decoder, reward_predictor, discriminator,
                      transition)
    rank, world = trainer.rank, trainer.world
    if args.batch_size % world:
        raise ValueError("--batch-size %d does not split evenly over %d ranks" % (args.batch_size, world))
    batch_size = args.batch_size // world
    ts = trainer.ts
    nets = [encoder, decoder, transition, discriminator, reward_predictor]
    stepper = trainer
    if getattr(args, 'graph_step', False):
        from .graph_step import GraphedTrainer
        stepper = GraphedTrainer(trainer)

    def horizon_of(it):                                                        # main.py:143-145
        return args.horizon_min + int((args.horizon_max - args.horizon_min) * (it / train_iters))
    queue = None
    if getattr(args, 'prefetch', 0) > 0:
        from .datasource import DeviceBatchQueue
        queue = DeviceBatchQueue(datasource, batch_size, [horizon_of(i) for i in range(args.start_iter, train_iters + 1)],
                                 num_actions, depth=args.prefetch)
    for train_iter in range(args.start_iter, train_iters + 1):
        if train_iter % args.eval_every == 0:
            # EVERY rank evaluates: SpectralNorm advances u / v on every forward, eval included
            # (spectral_normalization.py:66-68), and the iteration depends on the weights only -- so the
            # replicas stay bit-identical only if all of them make the same calls.  Rank 0 writes.
            evaluate(eval_datasource if eval_datasource is not None else datasource, encoder, decoder, transition,
                     discriminator, reward_predictor, latent_dim, train_iter=train_iter, quiet=rank != 0)
            if world > 1 and args.check_replicas:
                dp.assert_replicas_identical(nets, 'after the evaluation at iteration %d' % train_iter)
            if rank == 0:
                print('Saving networks to filesystem...')
                torch.save(transition.state_dict(), 'model-transition.pth')
                torch.save(encoder.state_dict(), 'model-encoder.pth')
                torch.save(decoder.state_dict(), 'model-decoder.pth')
                torch.save(discriminator.state_dict(), 'model-discriminator.pth')
                torch.save(reward_predictor.state_dict(), 'model-reward_predictor.pth')

        theta = train_iter / train_iters                                       # main.py:143-145
        pred_delta = args.horizon_max - args.horizon_min
        prediction_horizon = args.horizon_min + int(pred_delta * theta)

        if queue is not None:
            d_states, d_rewards, d_dones, d_onehot, actions = queue.get()
            stepper.step(d_states, d_rewards, d_dones, d_onehot, theta, train_iter, actions)
        else:
            states, rewards, dones, actions = datasource.get_trajectories(batch_size, prediction_horizon)
            stepper.step(_upload(states), _upload(rewards), _upload(np.asarray(dones).astype(int)),
                         onehot(actions, num_actions), theta, train_iter, actions)
        if world > 1 and train_iter % args.log_every == 0:
            ts.all_reduce_latest(world)        # the logged losses are local-batch means: average them over the ranks
        if rank == 0:
            ts.print_every(args.log_every)
    if rank == 0:
        print(ts)
        print('Finished')
    return ts


def evaluate(datasource, encoder, decoder, transition, discriminator, reward_predictor, latent_dim, train_iter=0,
             use_training_set=False, quiet=False):
    """main.py:315-322 (numeric part; the visualisers are out of scope).  quiet: compute only
    (ranks other than 0 of a data-parallel job), no prints, no files."""
    if not quiet:
        print('Evaluating networks...')
    test_mode([encoder, decoder, transition, discriminator, reward_predictor])
    return measure_prediction_mse(datasource, encoder, decoder, transition, reward_predictor, train_iter,
                                  num_factors=latent_dim, use_training_set=use_training_set, write_files=not quiet,
                                  quiet=quiet)


def compute_rollout_reward(z, transition, reward_predictor, num_actions,
                           selected_action, lookahead=2, rollout_depth=12,
                           rollout_policy='noop', negative_positive_tradeoff=10.0):
    """main.py:455-489: beam of num_actions**lookahead action sequences."""
    rollout_width = num_actions ** lookahead
    z = z.repeat(rollout_width, 1, 1, 1)
    actions = []
    for i in range(num_actions):
        for j in range(num_actions):
            if rollout_policy == 'noop':
                actions.append([i, j] + [0] * (rollout_depth - lookahead))
            elif rollout_policy == 'random':
                actions.append([i, j] + [np.random.randint(num_actions) for _ in range(rollout_depth - lookahead)])
    actions = np.array(actions)
    assert len(actions) == rollout_width
    cumulative_reward = reward_predictor(z).clone()
    for t in range(rollout_depth):
        z = z.detach()
        z = transition(z, onehot(actions[:, t], num_actions))
        cumulative_reward += reward_predictor(z)
    cumulative_reward[:, 0] *= negative_positive_tradeoff
    max_r, max_idx = cumulative_reward.sum(dim=1).max(dim=0)
    return max_r


@torch.no_grad()
def play(latent_dim, datasource, num_actions, num_rewards, encoder, decoder, reward_predictor, discriminator,
         transition, max_steps=300, use_graph=True, verbose=True):
    """main.py:327-400, the model-predictive agent (numeric part; the mp4 writer is out of scope):
    estimate the latent state from three frames, then every step roll every first action forward
    through compute_rollout_reward, act greedily in the real game and re-estimate the state from the
    last three frames.  use_graph replays each rollout as one HIP graph (planner.GraphedRollout).
    Returns (steps, cumulative_reward, [chosen actions])."""
    test_mode([encoder, decoder, transition, discriminator, reward_predictor])
    env = datasource.make_env(screen_size=512)
    no_op = min(3, num_actions - 1)                                    # the reference hard-codes button 3
    frames = [datasource.convert_frame(env.reset())[0]]
    done = False
    for _ in range(2):
        state, _, done, _ = env.step(no_op)
        frames.append(datasource.convert_frame(state)[0])

    def estimate(frames, action):
        z = encoder(_upload(np.stack(frames)).unsqueeze(0))
        return transition(z, onehot(int(action), num_actions))
    z = estimate(frames, no_op)
    planner = None
    if use_graph:
        from .planner import GraphedRollout
        planner = GraphedRollout(transition, reward_predictor, num_actions)
    t, total, pos, neg, chosen = 2, 0.0, 0.0, 0.0, []
    while not done:
        futures = []
        for a in range(num_actions):
            z_a = transition(z, onehot(a, num_actions))
            if planner is not None:
                futures.append(float(planner(z_a)))
            else:
                futures.append(float(compute_rollout_reward(z_a, transition, reward_predictor, num_actions, a,
                                                            rollout_depth=12, rollout_policy='noop')))
        best = int(np.argmax(futures))
        chosen.append(best)
        new_state, new_reward, done, info = env.step(best)
        if len(info) > 1:                                              # per-channel rewards (main.py:373-378)
            pos += sum(v for v in info.values() if v > 0)
            neg -= sum(v for v in info.values() if v < 0)
        else:
            pos += max(0, new_reward)
            neg -= min(0, new_reward)
        total += new_reward
        if verbose:
            print('t={} curr. r={:.02f} future r: {}'.format(t, total, ' '.join('{:.02f}'.format(f) for f in futures)))
            print('Negative Reward: {}    Positive Reward: {}'.format(int(neg), int(pos)))
        frames = frames[1:] + [datasource.convert_frame(new_state)[0]]
        z = estimate(frames, best)
        t += 1
        if t > max_steps:
            print('Ending evaluation due to time limit')
            break
    msg = 'Finished at t={} with cumulative reward {}'.format(t, total)
    with open('evaluation_metrics_{}.txt'.format(int(time.time())), 'w') as fp:
        fp.write(msg + '\n')
    print(msg)
    return t, total, chosen


@torch.no_grad()
def measure_prediction_mse(datasource, encoder, decoder, transition, reward_pred,
                           train_iter=0, timesteps=100, num_factors=16, experiment_name='default',
                           use_training_set=False, batch_size=100, write_files=True, quiet=False):
    """main.py:784-855: open-loop eval rollout, per-step masked pixel / reward MSE.
    Returns {'pixel': (means, stds), 'reward': (means, stds)} and, like the
    reference's plot_error_graph (main.py:858-865), writes the mse_*.json files.
    Runs under no_grad: the reference cuts the graph every step anyway (z.detach_(),
    main.py:835) and SpectralNorm advances u/v regardless of grad mode."""
    start_time = time.time()
    say = (lambda *a: None) if quiet else print
    num_actions = datasource.binary_input_channels
    states, rewards, dones, actions = datasource.get_trajectories(batch_size=batch_size, timesteps=timesteps,
                                                                  training=use_training_set)
    states = _upload(states)
    rewards = _upload(rewards)
    dones = _upload(np.asarray(dones).astype(int))
    onehots = onehot(actions, num_actions)

    z = encoder(states[:, :3])
    z = transition(z, onehots[:, 1])
    mse_losses, mse_stddevs, reward_losses, reward_stddevs = [], [], [], []
    active_mask = torch.ones(batch_size, device=_device())
    for t in range(2, timesteps):
        active_mask = active_mask * (1 - dones[:, t])
        if float(active_mask.sum()) == 0:
            say('Ending simulation at max trajectory length {}'.format(t))
            break
        diffs = active_mask * decoder_frame_loss(decoder, z, states[:, t].contiguous(), kind=1)
        scale = batch_size / torch.sum(active_mask)
        mse_losses.append(float(torch.mean(diffs) * scale))
        mse_stddevs.append(float(torch.std(diffs) * scale))
        r_expected = rewards[:, t].sum(-1)
        r_predicted = reward_pred(z).sum(-1)
        r_diffs = active_mask * (r_expected - r_predicted) ** 2
        reward_losses.append(float(torch.mean(r_diffs) * scale))
        reward_stddevs.append(float(torch.std(r_diffs) * scale))
        z = transition(z, onehots[:, t])
        z = z.detach()
    if len(mse_losses) == 0:
        say('Degenerate trajectory, skipping MSE calculation')
        return None
    say('MSE over {} timesteps: min {:.3f} max {:.3f}'.format(timesteps, min(mse_losses), max(mse_losses)))
    say(sparkline(mse_losses, length=80))
    say('Avg. MSE loss: {}'.format(np.mean(mse_losses)))
    say('Finished trajectory simulation in {:.02f}s'.format(time.time() - start_time))
    if write_files:
        for name, (m, s) in (('pixel', (mse_losses, mse_stddevs)), ('reward', (reward_losses, reward_stddevs))):
            with open('mse_{}_iter_{:06d}.json'.format(name, train_iter), 'w') as fp:
                fp.write(json.dumps(m, indent=2))
            with open('mse_stddev_{}_iter_{:06d}.json'.format(name, train_iter), 'w') as fp:
                fp.write(json.dumps(s, indent=2))
    return {'pixel': (mse_losses, mse_stddevs), 'reward': (reward_losses, reward_stddevs)}


def main(argv=None):
    """main.py:64-103."""
    global args
    args = parser.parse_args(argv)
    # one process per GPU: bind cuda:LOCAL_RANK and join the process group BEFORE any other GPU call
    rank, local_rank, world = dp.init_from_env()
    ops.set_compute_precision(args.precision)
    if args.conv_algo:
        ops.set_conv_algo(args.conv_algo)
    torch.manual_seed(args.seed)                   # identical initial replicas on every rank
    latent_dim = 16
    rank_seed = args.seed + dp.SEED_STRIDE * rank  # per-rank data stream and latent-noise stream
    datasource = allocate_datasource(args.env, seed=rank_seed)
    num_actions = datasource.binary_input_channels
    num_rewards = datasource.scalar_output_channels
    encoder = models.Encoder(latent_dim, datasource.conv_input_channels)
    decoder = models.Decoder(latent_dim, datasource.conv_output_channels)
    reward_predictor = models.RewardPredictor(latent_dim, num_rewards)
    discriminator = models.Discriminator()
    transition = models.Transition(latent_dim, num_actions)
    transition.seed_noise(rank_seed)
    if args.load_from is None:
        if rank == 0:
            print('No --load-from directory specified: initializing new networks')
    elif 'model-encoder.pth' not in os.listdir(args.load_from):
        raise ValueError('Failed to load weights from *.pth')
    else:
        print('Loading models from directory {}'.format(args.load_from))
        for net, name in ((encoder, 'encoder'), (decoder, 'decoder'), (transition, 'transition'),
                          (discriminator, 'discriminator'), (reward_predictor, 'reward_predictor')):
            net.load_state_dict(torch.load(os.path.join(args.load_from, 'model-{}.pth'.format(name)),
                                           weights_only=True))
    if world > 1:
        # whatever the ranks initialised or loaded, rank 0's parameters and u / v are THE replica
        dp.broadcast_replica([encoder, decoder, transition, discriminator, reward_predictor])
    if args.evaluate:
        # main.py:92-99: play-throughs with the model-predictive agent (datasources that can make_env), then the
        # open-loop prediction error
        try:
            datasource.make_env(screen_size=512)
            can_play = True
        except NotImplementedError:
            can_play = False
        if can_play:
            for _ in range(args.evaluations):
                play(latent_dim, datasource, num_actions, num_rewards, encoder, decoder, reward_predictor,
                     discriminator, transition, verbose=False)
            print('Finished {} playthroughs'.format(args.evaluations))
        evaluate(datasource, encoder, decoder, transition, discriminator, reward_predictor, latent_dim,
                 use_training_set=True)
    else:
        # every rank evaluates the SAME trajectories (see train()): a second datasource on the un-offset seed
        eval_ds = allocate_datasource(args.env, seed=args.seed + dp.SEED_STRIDE // 2) if world > 1 else None
        train(latent_dim, datasource, num_actions, num_rewards, encoder, decoder, reward_predictor, discriminator,
              transition, eval_datasource=eval_ds)
    if rank == 0:
        print('Finished execution, terminating')
    dp.shutdown()


if __name__ == '__main__':
    main()
