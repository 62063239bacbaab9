"""The training iteration as ONE HIP graph.

At small batches (BASELINE configs[0]: batch 8, horizon 5) an iteration is ~650 kernel launches of a
few microseconds each and the host cannot issue them as fast as the GPU retires them.  Every kernel of
libscmgan_hip.so is a plain launch on the caller's stream, so the whole of Trainer.step -- SpectralNorm
power iterations on the side stream, the unrolled rollout, autograd's backward, the fused clip + Adam --
captures into a HIP graph (torch.cuda.CUDAGraph drives hipStreamBeginCapture / hipGraphLaunch) and is
replayed per iteration.

What changes from iteration to iteration does not travel as kernel arguments (a graph freezes those) but
lives in device memory, read by the same kernels through optional pointers (include/scm_hip.h):
  * the batch: copied into static input buffers before the replay;
  * theta * reward_coef (main.py:143,185): ops.dev_scalars['coef'], written before the replay;
  * the Philox position of the latent sampler and the Adam step number: ops.dev_scalars['noise_off'] /
    ['adam_step'], written from the host-side counters before the replay.
Eager and captured iterations run the same kernels on the same values, so the weights after n iterations
are bitwise those of the eager loop (tests/test_gpu_graph_step.py).

One graph per (batch shape, horizon).  The first `warmup` iterations of a shape run eagerly -- they are
ordinary training iterations, nothing is executed twice; stream capture itself executes nothing.
Not captured (the eager step is used): optional loss terms that draw host-side random numbers
(counterfactual regularisers), latent overshooting, --gan, injected noise tensors, data-parallel runs.
"""
import torch

from . import ops


class _Recorder(object):
    """Stands in for TimeSeries while capturing: remembers (name, static tensor)."""

    def __init__(self):
        self.items = []

    def collect(self, name, value):
        self.items.append((name, value))


class GraphedTrainer(object):
    def __init__(self, trainer, warmup=2):
        a = trainer.a
        if trainer.world != 1:
            raise ValueError("GraphedTrainer: data-parallel iterations are not captured (RCCL stays eager)")
        if a.latent_overshooting or a.enable_disentanglement_loss or a.enable_action_control_loss or trainer.gan \
                or a.truncate_bptt:
            raise ValueError("GraphedTrainer captures the default loss only (no optional terms, no --gan)")
        self.trainer, self.warmup = trainer, warmup
        self._graphs = {}       # key -> dict(graph, inputs, loss, logged, noise_inc)
        self._seen = {}         # key -> eager iterations run so far
        dev = next(trainer.encoder.parameters()).device
        self._coef = torch.zeros(1, device=dev, dtype=torch.float32)
        self._noise_off = torch.zeros(1, device=dev, dtype=torch.int64)
        self._adam_step = torch.zeros(1, device=dev, dtype=torch.int32)
        self.replays = 0

    # ------------------------------------------------------------------ helpers
    def _optimizers(self):
        t = self.trainer
        opts = [t.opt_pred]
        if not t.a.finetune_reward:
            opts += [t.opt_enc, t.opt_dec, t.opt_trans]
        return opts

    def _python_step_counts(self):
        return [st["step"] for o in self._optimizers() for st in o.state.values()]

    def _host_counters(self, delta, noise_delta):
        """The host side of an iteration's bookkeeping (what the eager step does in Python)."""
        for o in self._optimizers():
            for st in o.state.values():
                st["step"] += delta
        self.trainer.transition._noise_offset += noise_delta

    def _capture(self, key, batch, theta, train_iter):
        t = self.trainer
        tr = t.transition
        if tr._noise_queue:
            raise ValueError("GraphedTrainer: injected noise tensors cannot be captured (use the Philox stream)")
        if len(set(self._python_step_counts())) != 1:
            raise ValueError("GraphedTrainer: the optimizers must have taken the same number of steps")
        inputs = [b.clone() for b in batch]
        rec = _Recorder()
        real_ts, t.ts = t.ts, rec
        noise_base = tr._noise_offset
        ops.dev_scalars.update(coef=self._coef, noise_off=self._noise_off, adam_step=self._adam_step)
        # Every weight pack and SpectralNorm stamp must be REBUILT INSIDE the capture: a pack cached by an eager
        # forward since the last optimizer step (an evaluate() on the capture iteration) would otherwise be hit
        # during capture, no pack kernel would be recorded, and every replay would convolve with capture-time
        # weights while the in-graph Adam keeps updating the real ones.
        ops.bump_weights_epoch()
        graph = torch.cuda.CUDAGraph()
        try:
            torch.cuda.synchronize()
            with torch.cuda.graph(graph):
                loss = t.step(inputs[0], inputs[1], inputs[2], inputs[3], theta, train_iter)
        finally:
            ops.dev_scalars.clear()
            t.ts = real_ts
        # Stream capture executed nothing on the device, but the host side of one iteration ran (python step
        # counters, noise offset): take that back, every replay then does its own bookkeeping.
        noise_inc = tr._noise_offset - noise_base
        self._host_counters(-1, -noise_inc)
        g = {"graph": graph, "inputs": inputs, "loss": loss, "logged": rec.items, "noise_base": noise_base,
             "noise_inc": noise_inc}
        self._graphs[key] = g
        return g

    # --------------------------------------------------------------------- step
    def step(self, states, rewards, dones, onehots, theta, train_iter=1, actions=None):
        """Same contract as Trainer.step (device fp32 tensors); returns the (device) loss."""
        t = self.trainer
        batch = (states, rewards, dones, onehots)
        key = tuple(tuple(b.shape) for b in batch)
        g = self._graphs.get(key)
        if g is None:
            n = self._seen.get(key, 0)
            if n < self.warmup:
                self._seen[key] = n + 1
                return t.step(states, rewards, dones, onehots, theta, train_iter, actions)
            g = self._capture(key, batch, theta, train_iter)
        for dst, src in zip(g["inputs"], batch):
            dst.copy_(src)
        # this iteration's scalars: theta * reward_coef, the sampler's stream position relative to the offsets
        # baked at capture, the Adam step number
        self._coef.fill_(theta * t.a.reward_coef)
        self._noise_off.fill_(t.transition._noise_offset - g["noise_base"])
        self._adam_step.fill_(self._python_step_counts()[0] + 1)
        g["graph"].replay()
        self.replays += 1
        self._host_counters(+1, g["noise_inc"])
        ops.bump_weights_epoch()        # the graph updated the weights through raw pointers: packs cached by eager code are stale
        if g["logged"]:
            vals = torch.stack([v.detach().reshape(()) for _, v in g["logged"]])     # one copy out of the static buffers
            for i, (name, _) in enumerate(g["logged"]):
                t.ts.collect(name, vals[i])
        return g["loss"].clone()
