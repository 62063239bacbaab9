"""Minimal stand-in for the `logutil.TimeSeries` the reference logs through
(main.py:130,184,196,297-298; the real package is a git dependency that is not
available offline).  Values stay on the device until print time, so collecting
a loss never forces a host sync inside the step."""
import time


class TimeSeries(object):
    def __init__(self, title='', epoch_length=None, tensorboard=False):
        self.title = title
        self.epoch_length = epoch_length
        self.series = {}
        self.counter = 0
        self.start = time.time()

    def collect(self, name, value):
        self.series.setdefault(name, []).append(value.detach() if hasattr(value, 'detach') else value)

    def all_reduce_latest(self, world):
        """Data parallel: replace the latest value of every series by its mean over the ranks
        (one small all-reduce at print time; every rank must call it with the same series)."""
        import torch
        import torch.distributed as dist
        names = sorted(k for k, v in self.series.items() if v and hasattr(v[-1], 'detach'))
        if not names:
            return
        packed = torch.stack([self.series[k][-1].detach().reshape(()).float() for k in names])
        dist.all_reduce(packed, op=dist.ReduceOp.SUM)
        packed = packed / world
        for i, k in enumerate(names):
            self.series[k][-1] = packed[i]

    def _latest(self):
        return {k: float(v[-1]) for k, v in self.series.items() if v}

    def print_every(self, n):
        self.counter += 1
        if self.counter % n == 0:
            print(self.format_all())

    def format_all(self):
        head = '[{} {}/{} {:.1f}s]'.format(self.title, self.counter, self.epoch_length, time.time() - self.start)
        return head + ' ' + ' '.join('{}={:.4f}'.format(k, v) for k, v in sorted(self._latest().items()))

    def values(self, name):
        return [float(v) for v in self.series.get(name, [])]

    def __str__(self):
        return self.format_all()


def sparkline(values, length=80):
    bars = u' ▁▂▃▄▅▆▇█'
    if not values:
        return ''
    lo, hi = min(values), max(values)
    span = (hi - lo) or 1.0
    step = max(1, len(values) // length)
    return ''.join(bars[int((v - lo) / span * (len(bars) - 1))] for v in values[::step])
