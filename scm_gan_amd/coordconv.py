"""CoordConv2d with the reference's interface (coordconv.py:5-15): appends x / y
coordinate planes in [-1, 1) and applies a Conv2d built from the caller's
arguments (the caller passes in_channels + 2, as in the reference)."""
from torch import nn

from . import ops
from .layers import Conv2d


class CoordConv2d(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        self.conv = Conv2d(*args, **kwargs)

    def forward(self, x):
        batch_size, channels, height, width = x.shape
        if height != width:
            # the reference builds coord_x with repeat(width, 1) and coord_y with
            # repeat(1, height): torch.cat fails for non-square inputs
            raise RuntimeError("CoordConv2d: square inputs only (reference coordconv.py:12-14)")
        return self.conv(ops.CoordCatFn.apply(x))
