"""CSRN (convolutional spatial-recurrent network) operator, reference
spatial_recurrent.py:21-119.  HIP kernels: csrc/csrn.hip."""
import torch
from torch import nn

from . import ops


class CSRN(nn.Module):
    def __init__(self, channels):
        super().__init__()
        raise NotImplementedError("CSRN HIP operator: scheduled after the main path (see DESIGN.md)")
