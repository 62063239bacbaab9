"""SpectralNorm wrapper with the reference's interface and state layout.

Mirrors /root/reference/spectral_normalization.py:14-68: wraps a module, removes
its `weight` parameter and registers `weight_u`, `weight_v` (requires_grad=False)
and `weight_bar` ON THE WRAPPED MODULE, runs ONE power iteration on EVERY forward
call (train, eval and no_grad alike -- SURVEY.md section 0.6) and normalises the
weight by sigma = u . (W_bar v).  The power iteration, the 1/sigma scaling (in
the conv epilogue) and the gradient through sigma are HIP kernels
(csrc/sn.hip); W_bar / sigma is never materialised.
"""
import torch
from torch import nn
from torch.nn import Parameter

from . import ops


def l2normalize(v, eps=1e-12):
    return v / (v.norm() + eps)


class SpectralNorm(nn.Module):
    """`SpectralNorm(module, name='weight', power_iterations=1)`: same constructor, same
    parameters registered on the wrapped module (`<name>_u`, `<name>_v`, `<name>_bar`)."""

    SUFFIXES = ("_u", "_v", "_bar")

    def __init__(self, module, name='weight', power_iterations=1):
        super(SpectralNorm, self).__init__()
        self.module, self.name, self.power_iterations = module, name, power_iterations
        self._precomputed = []      # (sigma, inv_sigma) of power iterations already run (see precompute())
        if not all(hasattr(module, name + s) for s in self.SUFFIXES):
            self._install_parameters()

    def _install_parameters(self):
        """Replace `weight` by `weight_bar` plus the two power-iteration vectors
        (u: rows, v: columns of the weight viewed as a matrix), unit-normalised draws
        from N(0, 1) that never receive gradients -- spectral_normalization.py:47-63."""
        weight = getattr(self.module, self.name)
        rows = weight.shape[0]
        cols = weight.numel() // rows
        vectors = {}
        for suffix, n in (("_u", rows), ("_v", cols)):
            vec = weight.data.new_empty(n).normal_(0, 1)
            vectors[suffix] = Parameter(l2normalize(vec), requires_grad=False)
        self.module._parameters.pop(self.name)
        self.module.register_parameter(self.name + "_u", vectors["_u"])
        self.module.register_parameter(self.name + "_v", vectors["_v"])
        self.module.register_parameter(self.name + "_bar", Parameter(weight.data))

    def _update_u_v(self):
        """Power iteration(s) on the device; returns (sigma, inv_sigma)."""
        u = getattr(self.module, self.name + "_u")
        v = getattr(self.module, self.name + "_v")
        w = getattr(self.module, self.name + "_bar")
        sigma = inv_sigma = None
        if self._precomputed:
            stamp, sigma, inv_sigma = self._precomputed.pop(0)
            if stamp == _stamp(w):
                return w, u, v, sigma, inv_sigma
            # W_bar changed (optimizer step, load_state_dict) since precompute(): the queued sigmas belong
            # to the old weights.  u / v have already advanced past them; drop the rest and iterate live.
            self._precomputed = []
        for _ in range(self.power_iterations):
            sigma, inv_sigma = ops.sn_power_iter(w, u, v)
        return w, u, v, sigma, inv_sigma

    def forward(self, *args, **kwargs):
        w, u, v, sigma, inv_sigma = self._update_u_v()
        if self.name != 'weight' or not hasattr(self.module, "forward") or \
                "_sn" not in self.module.forward.__code__.co_varnames:
            raise TypeError("scm_gan_amd.SpectralNorm wraps scm_gan_amd.layers modules (Conv2d); got %r"
                            % type(self.module).__name__)
        return self.module.forward(*args, _sn=(w, u, v, inv_sigma), **kwargs)


def _stamp(w):
    """Identity of the W_bar a queued sigma was computed from."""
    return (ops.weights_epoch(), w._version, w.data_ptr())


def _spectral_norms(modules):
    sns = []
    for m in modules:
        sns += [x for x in m.modules() if isinstance(x, SpectralNorm)]
    return sns


def clear_precomputed(modules):
    """Drop sigmas queued by precompute() and not consumed (an iteration that raised midway)."""
    for x in _spectral_norms(modules):
        x._precomputed = []


def precompute(modules, calls):
    """Run the power iterations of the next `calls` forward calls of every
    SpectralNorm found in `modules`, batched: the iteration depends only on
    (W_bar, u, v), never on the data, so all of an unrolled rollout's iterations
    can run ahead of it in one launch per 8 layers (u, v end in the same state and
    every call still sees its own sigma).  Results are queued per module and
    consumed by the following forward calls; each entry carries the identity of the W_bar
    it was computed from and is dropped when that no longer matches."""
    sns = [x for x in _spectral_norms(modules) if x.power_iterations == 1 and x.name == 'weight']
    for i in range(0, len(sns), 8):
        group = sns[i:i + 8]
        layers = [(getattr(x.module, "weight_bar"), getattr(x.module, "weight_u"), getattr(x.module, "weight_v"))
                  for x in group]
        out = ops.sn_power_iter_multi(layers, calls)
        for k, x in enumerate(group):
            stamp = _stamp(layers[k][0])
            x._precomputed += [(stamp, out[k, 0, c:c + 1], out[k, 1, c:c + 1]) for c in range(calls)]
    return len(sns)
