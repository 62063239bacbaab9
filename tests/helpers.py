"""Shared helpers for the parity tests."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def digest(t, n=4096):
    import weights as WS
    return WS.digest(t, n)


def to_np(x):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.asarray(x)


def check(fx, key, actual, atol, rtol=0.0, what=""):
    """Compare `actual` with fixture entry `key`; entries stored as a digest
    (key#digest: moments + strided sample, see weights.digest) are compared
    digest-to-digest with the moments scaled by sqrt(size)."""
    a = to_np(actual)
    if key in fx.files:
        e = fx[key]
        assert a.shape == e.shape, "%s %s: shape %s vs %s" % (what, key, a.shape, e.shape)
        err = np.abs(a.astype(np.float64) - e.astype(np.float64))
        tol = atol + rtol * np.abs(e)
        assert np.all(err <= tol), "%s %s: max err %.3e (tol %.1e) at %s" % (
            what, key, err.max(), atol, np.unravel_index(err.argmax(), err.shape))
        return float(err.max())
    dk = key + "#digest"
    assert dk in fx.files, "fixture has no entry %s" % key
    assert tuple(fx[key + "#shape"]) == tuple(a.shape), "%s %s: shape" % (what, key)
    e = fx[dk]
    d = digest(a, 4096)
    err = np.abs(d[3:] - e[3:])
    tol = atol + rtol * np.abs(e[3:])
    assert np.all(err <= tol), "%s %s: sample max err %.3e (tol %.1e)" % (what, key, err.max(), atol)
    # moments: sum / abs-sum are sums of `size` terms, L2 is a norm
    n = a.size
    assert abs(d[0] - e[0]) <= (atol + rtol * abs(e[1]) / n) * n, "%s %s: sum %.6e vs %.6e" % (what, key, d[0], e[0])
    assert abs(d[2] - e[2]) <= atol * np.sqrt(n) + rtol * e[2], "%s %s: L2 %.6e vs %.6e" % (what, key, d[2], e[2])
    return float(err.max())
