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
simulators that need gym_minipacman, StarCraft II or ALE are out of scope and not
available offline; SyntheticDatasource produces batches of the same shapes and
dtypes from a seeded numpy stream.  Its three pure-numpy
games ('pong', 'gridworld', 'gameoflife') are restated in simulators.py.
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
    if datasource_name in ("pong", "gridworld", "gameoflife"):
        # the reference's pure-numpy simulators, restated as batch datasources (simulators.py)
        from . import simulators
        return {"pong": simulators.Pong, "gridworld": simulators.GridWorld,
                "gameoflife": simulators.GameOfLife}[datasource_name](seed=seed)
    if datasource_name in _REFERENCE_ENVS:
        raise ValueError(
            "datasource '%s' needs the reference's simulator stack (gym / gym_minipacman / sc2env), which is out "
            "of scope for this build; use 'synthetic_%s'-style names or pass any object with the "
            "get_trajectories contract" % (datasource_name, datasource_name))
    raise ValueError('Failed to find datasource with name {}'.format(datasource_name))


class DeviceBatchQueue(object):
    """Device-side batch interface: replaces the per-iteration
    `torch.Tensor(numpy).cuda()` + host one-hot of the reference loop (main.py:156-158, 206).

    A worker thread draws trajectories from the datasource in iteration order, converts them to
    fp32 in pinned host buffers (two slots) and uploads them with asynchronous copies on a side
    HIP stream; the one-hot action tensor is built on the device (scm_onehot).  get() hands the
    device tensors of the next iteration to the training stream behind an event wait (no host
    synchronisation).  The datasource sees exactly the calls of the serial loop, in the same
    order, so the batches are identical to it; a datasource whose training and evaluation draws
    share one random stream would see them interleaved differently, which is why train() only
    uses the queue on request (--prefetch).
    """

    def __init__(self, datasource, batch_size, horizons, num_actions, depth=2):
        import queue
        import threading
        import torch
        self._torch = torch
        self.ds, self.B, self.A = datasource, batch_size, num_actions
        self.horizons = list(horizons)              # horizon of every iteration, in order
        self._q = queue.Queue(maxsize=max(1, depth))
        self._dev = torch.device('cuda', torch.cuda.current_device())
        self._stream = torch.cuda.Stream(device=self._dev)
        self._pinned = [None, None]                 # per slot: dict name -> pinned tensor
        self._copied = [None, None]                 # per slot: event "the uploads of this slot are done"
        self._err = None
        self._thread = threading.Thread(target=self._work, daemon=True)
        self._thread.start()

    def _pin(self, slot, name, arr, dtype):
        torch = self._torch
        cur = self._pinned[slot] = self._pinned[slot] or {}
        t = cur.get(name)
        if t is None or t.shape != arr.shape or t.dtype != dtype:
            t = cur[name] = torch.empty(arr.shape, dtype=dtype, pin_memory=True)
        t.numpy()[...] = arr                        # converting copy (e.g. float64 -> float32) straight into pinned memory
        return t

    def _work(self):
        torch = self._torch
        from . import _lib as L
        import ctypes
        try:
            torch.cuda.set_device(self._dev)
            for it, h in enumerate(self.horizons):
                states, rewards, dones, actions = self.ds.get_trajectories(self.B, h)[:4]
                slot = it & 1
                if self._copied[slot] is not None:
                    self._copied[slot].synchronize()          # the pinned slot is free again
                hs = self._pin(slot, 's', np.asarray(states), torch.float32)
                hr = self._pin(slot, 'r', np.asarray(rewards), torch.float32)
                hd = self._pin(slot, 'd', np.asarray(dones).astype(np.float32), torch.float32)
                ha = self._pin(slot, 'a', np.asarray(actions).astype(np.int64), torch.int64)
                with torch.cuda.stream(self._stream):
                    ds_, dr, dd, da = (t.to(self._dev, non_blocking=True) for t in (hs, hr, hd, ha))
                    oh = torch.empty(tuple(da.shape) + (self.A,), device=self._dev, dtype=torch.float32)
                    L.check(L.lib().scm_onehot(ctypes.c_void_p(da.data_ptr()), da.numel(), self.A,
                                               ctypes.c_void_p(oh.data_ptr()),
                                               ctypes.c_void_p(self._stream.cuda_stream)), "scm_onehot")
                    ev = torch.cuda.Event()
                    ev.record(self._stream)
                self._copied[slot] = ev
                self._q.put((ds_, dr, dd, oh, np.asarray(actions), ev))
            self._q.put(None)
        except BaseException as e:                  # surfaced by get()
            self._err = e
            self._q.put(None)

    def get(self):
        """(states, rewards, dones, onehot_actions, numpy_actions) of the next iteration, device tensors
        valid on the current stream."""
        item = self._q.get()
        if item is None:
            if self._err is not None:
                raise self._err
            raise StopIteration("DeviceBatchQueue: all %d batches consumed" % len(self.horizons))
        s, r, d, oh, a, ev = item
        cur = self._torch.cuda.current_stream()
        cur.wait_event(ev)
        for t in (s, r, d, oh):
            t.record_stream(cur)                    # allocated on the side stream, consumed on this one
        return s, r, d, oh, a
