"""Excitation backprop-through-time over the learned model: numeric part of the
reference's excitation_bptt.visualize_bptt (/root/reference/excitation_bptt.py:6-53).

Rolls the transition forward keeping gradients on every latent z_t and every
one-hot action a_t; once the predicted reward magnitude crosses the threshold it
backpropagates from the arg-max pixel of the reward map and returns, per step,
the action attributions a_t * da_t and the latent masks |dz_t| / (.001 + max|dz_t|).
Video output (imutil.Video) is out of scope; callers get the tensors.
"""
import torch

from .main import onehot


def excitation_backtrack(z, transition, reward_predictor, num_actions, steps=30, threshold=0.8,
                         first_action=1, later_action=3):
    """Returns None if no step reaches the reward threshold, else a dict with
    't' (step index), 'reward' (B, R), 'action_attributions' (list of (1, A)) and
    'latent_masks' (list of masks like z, latest step first -- the reference's
    'Causal Backtrack' order, excitation_bptt.py:44-49)."""
    z = z.detach().requires_grad_(True)
    actions, zees = [], []
    for t in range(steps):
        a = onehot(first_action if t == 0 else later_action, num_actions).clone().requires_grad_(True)
        actions.append(a)
        z = transition(z, a)
        z.retain_grad()
        zees.append(z)
        r, rmap = reward_predictor(z, visualize=True)
        if float(r.detach().sum().abs()) > threshold:
            peak = (rmap.abs() == rmap.abs().max()).to(rmap.dtype)      # excitation_bptt.py:36
            (rmap * peak).sum().backward()
            return {
                "t": t,
                "reward": r.detach(),
                "action_attributions": [(at * at.grad).detach() for at in actions],
                "latent_masks": [(zt.grad.abs() / (.001 + zt.grad.abs().max())).detach() for zt in zees[::-1]],
            }
    return None
