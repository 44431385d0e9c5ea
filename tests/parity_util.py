"""Helpers shared by the parity tests."""
import numpy as np


def assert_rows_match(a, b, atol, what='rows'):
    """a, b: [n, d].  Every row of ``b`` must have a distinct partner row in
    ``a`` within ``atol`` (max-abs): equal up to a permutation.  Used for
    top-k outputs, where near-tied scores may swap neighbours."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    used = np.zeros(len(a), dtype=bool)
    for i in range(len(b)):
        d = np.abs(a - b[i]).max(axis=1)
        d[used] = np.inf
        j = int(np.argmin(d))
        assert d[j] <= atol, '%s: row %d has no partner (best %.3g)' % (
            what, i, d[j])
        used[j] = True


def frac_within(a, b, atol):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float((np.abs(a - b) <= atol).mean())
