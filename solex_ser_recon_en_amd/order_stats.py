"""np.percentile / np.median as order statistics: the GPU selects the two neighbouring order statistics (radix select),
the host interpolates them the way NumPy's _quantile / _lerp do (solex_util.py:535-537, 591-592; ellipse_to_circle.py:165,
241).  (The limb detection and the ellipse fit of ellipse_to_circle live in csrc/stages.hip and csrc/hostmath.hip.)"""
import math

import numpy as np


# ---- np.percentile / np.median as order statistics (the GPU selects, the host interpolates) -------------------
def lerp_order_stats(n, q):
    """np.percentile(.., q) (method 'linear') on n values = lerp between two order statistics:
    returns (rank_lo, rank_hi, combine(a, b)), following NumPy's _quantile / _lerp."""
    virtual = (n - 1) * np.true_divide(q, 100)
    lo = min(max(math.floor(virtual), 0), n - 1)
    hi = min(lo + 1, n - 1)
    gamma = virtual - math.floor(virtual)

    def combine(a, b):
        diff = b - a
        return b - diff * (1 - gamma) if gamma >= 0.5 else a + diff * gamma
    return lo, hi, combine


def lerp_gamma(n, q):
    """The interpolation weight lerp_order_stats' combine() uses."""
    virtual = (n - 1) * np.true_divide(q, 100)
    return virtual - math.floor(virtual)


def median_order_stats(n):
    """np.median on n values: the middle order statistic, or the mean of the two middle ones."""
    if n % 2:
        return n // 2, n // 2, (lambda a, b: a)
    return n // 2 - 1, n // 2, (lambda a, b: (a + b) / 2)
