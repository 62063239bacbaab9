"""clip_grad_value_ + Adam in one multi-tensor HIP launch per parameter group.

Semantics of torch.optim.Adam(params, lr) with default betas (0.9, 0.999),
eps 1e-8, no weight decay, preceded (optionally) by
torch.nn.utils.clip_grad_value_(params, clip) -- reference main.py:125-129,
287-296.  Parameters whose .grad is None are skipped, as torch does (the
Encoder's unused bn_conv1 and every SpectralNorm u/v).
"""
import ctypes

import torch

from . import _lib as L
from . import ops


class AdamClip(object):
    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, clip_value=0.0):
        self.params = [p for p in params]
        self.lr, self.betas, self.eps, self.clip_value = lr, betas, eps, clip_value
        self.state = {}
        self.grad_scale = 1.0

    def zero_grad(self, set_to_none=False):
        for p in self.params:
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.detach_()
                    p.grad.zero_()

    def step(self):
        live = [p for p in self.params if p.grad is not None]
        if not live:
            return
        groups = {}
        for p in live:
            st = self.state.get(id(p))
            if st is None:
                st = {"step": 0, "m": torch.zeros_like(p.data), "v": torch.zeros_like(p.data)}
                self.state[id(p)] = st
            st["step"] += 1
            groups.setdefault(st["step"], []).append((p, st))
        lib = L.lib()
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for step, items in groups.items():
            n = len(items)
            arr = ctypes.c_void_p * n
            ps = arr(*[p.data.data_ptr() for p, _ in items])
            gs = arr(*[p.grad.data_ptr() for p, _ in items])
            ms = arr(*[st["m"].data_ptr() for _, st in items])
            vs = arr(*[st["v"].data_ptr() for _, st in items])
            for p, _ in items:
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()):
                    raise RuntimeError("AdamClip: parameters and grads must be contiguous fp32 GPU tensors")
            ns = (ctypes.c_int * n)(*[p.numel() for p, _ in items])
            step_dev = ops.dev_scalars.get("adam_step")     # captured iteration: the step number lives on the device
            L.check(lib.scm_adam_clip_multi(n, ps, gs, ms, vs, ns, self.lr, self.betas[0], self.betas[1], self.eps,
                                            step, None if step_dev is None else ctypes.c_void_p(step_dev.data_ptr()),
                                            self.clip_value, self.grad_scale, stream), "scm_adam_clip_multi")
        ops.bump_weights_epoch()
