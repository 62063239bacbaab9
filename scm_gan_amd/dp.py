"""Data-parallel plumbing of the training iteration.

The reference is single-process (main.py:125-129, 285-296); the north star adds: one
process per GPU, the minibatch sharded over the ranks, gradients summed with RCCL
(`torch.distributed` backend "nccl" on ROCm) before the clip, overlapped with the tail
of the backward pass.  What lives here is host logic only:

  * init_from_env()     RANK / LOCAL_RANK / WORLD_SIZE -> device + process group;
  * FlatGrads           the gradients of one network as views of one flat fp32 bucket
                        (one all-reduce per network, no pack / unpack kernels);
  * GradSync            issues the all-reduces of the buckets that are complete when the
                        gradient of the first encoder output arrives (transition,
                        decoder, reward head, discriminator) on a side stream from an
                        autograd hook, so they run under the encoder's backward
                        convolutions; the encoder's bucket follows after backward();
                        clip + Adam wait on all of them;
  * broadcast_replica() / replica_fingerprint()   identical replicas at start, and a
                        cheap check that they stayed identical.

N ranks with batch B/N each reproduce one rank with batch B: every loss is a mean over
the local batch (main.py:183,195), so the full-batch gradient is the mean of the
per-rank gradients = SUM all-reduce, then the 1/world factor applied by the optimizer
(AdamClip.grad_scale) BEFORE the clip (main.py:287-290 clips the full-batch gradient).
"""
import os

import torch

SEED_STRIDE = 1000003      # rank r draws its data and its latent noise from seed + r * SEED_STRIDE


def env_world():
    """(rank, local_rank, world_size, under_launcher) from the environment torch.distributed.run sets."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    return rank, local_rank, world, ("RANK" in os.environ and "MASTER_PORT" in os.environ)


def init_from_env(backend=None):
    """Bind this process to its GPU and join the process group when it was started by a
    launcher (torch.distributed.run exports RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).
    Must run before anything else touches the GPU.  Returns (rank, local_rank, world)."""
    rank, local_rank, world, launched = env_world()
    if not launched:
        if world > 1:
            raise SystemExit("WORLD_SIZE=%d but RANK / MASTER_PORT are not set: start the ranks with "
                             "`python -m torch.distributed.run --nproc-per-node %d ...`" % (world, world))
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank)
        return 0, local_rank, 1
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.device_count() > 0 else "gloo"
    if backend == "nccl":
        have = torch.cuda.device_count()
        if local_rank >= have:
            raise SystemExit("rank %d (local rank %d of a %d-rank job) needs %d GPUs on this node, found %d"
                             % (rank, local_rank, world, local_rank + 1, have))
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend=backend)
    return rank, local_rank, world


def world_info():
    """(rank, world) of the initialised process group, (0, 1) without one."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shutdown():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


class FlatGrads(object):
    """Gradients of one network as views of a single flat fp32 buffer: one all-reduce per
    network, and autograd / the HIP weight-gradient kernels accumulate straight into it."""

    def __init__(self, module, name=""):
        self.name = name
        self.params = [p for p in module.parameters() if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, device=self.params[0].device, dtype=torch.float32)
        o = 0
        for p in self.params:
            p.grad = self.flat[o:o + p.numel()].view_as(p)
            o += p.numel()

    def zero(self):
        self.flat.zero_()
        lo, hi = self.flat.data_ptr(), self.flat.data_ptr() + 4 * self.flat.numel()
        for p in self.params:       # the views must still be the .grad tensors
            if p.grad is None or not (lo <= p.grad.data_ptr() < hi):
                raise RuntimeError("flat gradient view of network '%s' was replaced" % self.name)

    def all_reduce(self, async_op=False):
        import torch.distributed as dist
        return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=async_op)


class GradSync(object):
    """SUM all-reduce of the flat buckets of one iteration, overlapped with backward.

    arm(z) registers a hook on the first encoder output.  The autograd engine runs the
    nodes of a single-device graph in reverse creation order, so when the gradient of z
    arrives every node created after the first encoder call -- all transition, decoder,
    reward-head (and discriminator) work of the iteration -- has been enqueued on the
    training stream and only the encoder's own backward remains.  The hook makes the
    communication stream wait for that point and issues the `early` buckets' all-reduces
    there; they run on RCCL's stream under the encoder's backward convolutions.
    finish() issues the `late` buckets (the encoder's) and makes the training stream
    wait for every collective: clip + Adam come after it.
    """

    def __init__(self, early, late, enabled=True):
        self.early, self.late, self.enabled = list(early), list(late), enabled
        self._works = []
        self._early_issued = False
        self._main = None
        self._comm = torch.cuda.Stream() if enabled else None
        self.hook_fired = 0          # diagnostics / tests: how many iterations issued early
        # diagnostics (bench.py `allreduce_us`): with `timing` on, every issue is bracketed by HIP events on the
        # communication stream, which then itself waits for the collectives (RCCL runs them on its own stream)
        self.timing = False
        self.timings = []            # (tag, start event, end event)

    def arm(self, z):
        self._works, self._early_issued = [], False
        if not self.enabled or not self.early or not z.requires_grad:
            return
        self._main = torch.cuda.current_stream()
        z.register_hook(self._on_first_latent_grad)

    def _issue(self, buckets, tag):
        self._comm.wait_stream(self._main)          # everything enqueued so far on the training stream
        with torch.cuda.stream(self._comm):
            if self.timing:
                start = torch.cuda.Event(enable_timing=True)
                start.record()
            works = [b.all_reduce(async_op=True) for b in buckets]
            if self.timing:
                for w in works:
                    w.wait()                        # stream-side: the communication stream waits for RCCL's
                end = torch.cuda.Event(enable_timing=True)
                end.record()
                self.timings.append((tag, start, end))
            self._works += works

    def timing_summary(self):
        """{'early' / 'late' / 'all': mean microseconds between issue and completion on the communication stream}."""
        torch.cuda.synchronize()
        acc = {}
        for tag, s, e in self.timings:
            acc.setdefault(tag, []).append(s.elapsed_time(e) * 1e3)
        return {k: sum(v) / len(v) for k, v in acc.items()}

    def _on_first_latent_grad(self, grad):
        if not self._early_issued:
            self._early_issued = True
            self.hook_fired += 1
            self._issue(self.early, "early")
        return None

    def finish(self):
        """After backward(): all remaining buckets, then wait (stream-side) for all of them."""
        if not self.enabled:
            return
        if self._main is None:
            self._main = torch.cuda.current_stream()
        if self._early_issued:
            self._issue(self.late, "late")
        else:
            self._issue(self.early + self.late, "all")
        for w in self._works:
            w.wait()
        self._works = []


def replica_tensors(modules):
    """Everything a replica consists of: parameters (incl. SpectralNorm u / v) and buffers."""
    out = []
    for m in modules:
        out += [p.data for p in m.parameters()]
        out += [b for b in m.buffers() if b.dtype.is_floating_point]
    return out


def broadcast_replica(modules, src=0):
    """One-time broadcast of rank `src`'s parameters, u / v and buffers."""
    import torch.distributed as dist
    for t in replica_tensors(modules):
        dist.broadcast(t, src=src)


def replica_fingerprint(modules):
    """(sum, sum of squares) over every replica tensor, as a 2-element float64 tensor."""
    s = None
    for t in replica_tensors(modules):
        d = t.detach().double()
        v = torch.stack([d.sum(), (d * d).sum()])
        s = v if s is None else s + v
    return s


def assert_replicas_identical(modules, what=""):
    """Raises unless every rank holds bit-identical parameters and u / v (their fingerprints agree)."""
    import torch.distributed as dist
    f = replica_fingerprint(modules)
    lo, hi = f.clone(), f.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if not torch.equal(lo, hi):
        raise RuntimeError("data-parallel replicas diverged %s: fingerprint min %s max %s"
                           % (what, lo.tolist(), hi.tolist()))
