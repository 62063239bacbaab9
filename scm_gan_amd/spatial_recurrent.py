"""CSRN (convolutional spatial-recurrent network) with the reference's interface
and state_dict keys (/root/reference/spatial_recurrent.py:21-119).  The four
directional GRU + Conv1d sweeps run in csrc/csrn.hip, the 1x1 combine on the
direct conv kernel.  nn.Conv1d / nn.GRU objects are kept only as parameter
holders (names, shapes, initialisation); their forward is never called."""
from torch import nn

from . import ops
from .layers import Conv2d


def init_conv_weight(conv, channels):
    # spatial_recurrent.py:11-13: N(0, channels) -- "the RNN gradient should explode, not vanish"
    conv.weight.data.normal_(0, channels)


def init_gru_weight(gru, channels):
    # spatial_recurrent.py:15-18
    gru.weight_hh_l0.data.normal_(0, channels)
    gru.weight_ih_l0.data.normal_(0, channels)


class CSRN(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.channels = channels
        self.rnn_in = channels
        self.rnn_out = self.rnn_in
        for d in ("down", "up", "left", "right"):
            setattr(self, "conv_" + d, nn.Conv1d(self.rnn_out, self.rnn_in, kernel_size=3, stride=1, padding=1))
        for d in ("down", "up", "left", "right"):
            setattr(self, "rnn_" + d, nn.GRU(self.rnn_in, self.rnn_out, bias=False))
        self.conv_combine = Conv2d(self.rnn_in * 4, channels, kernel_size=1)
        for d in ("down", "up", "left", "right"):
            init_conv_weight(getattr(self, "conv_" + d), channels)
        for d in ("down", "up", "left", "right"):
            init_gru_weight(getattr(self, "rnn_" + d), channels)

    def forward(self, x):
        batch_size, channels, height, width = x.shape
        assert channels == self.channels
        params = []
        for d in ("down", "up", "left", "right"):
            rnn, conv = getattr(self, "rnn_" + d), getattr(self, "conv_" + d)
            params += [rnn.weight_ih_l0, rnn.weight_hh_l0, conv.weight, conv.bias]
        context_map = ops.CSRNSweepsFn.apply(x, *params)
        return self.conv_combine(context_map)
