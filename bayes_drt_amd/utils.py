"""Grid utilities on the hot path (reference bayes_drt/utils.py:113-140)."""
import numpy as np


def rel_round(x, precision):
    """Round to `precision` significant digits (the reference compares grids after this rounding)."""
    x_arr = np.asarray(x, dtype=float)
    mag = np.floor(np.log10(x_arr + 1e-30))       # 1e-30 guards zeros, as the reference does
    nd = (precision - mag).astype(int)
    if x_arr.ndim == 0:
        return round(float(x_arr), int(nd))
    return np.array([round(float(v), int(d)) for v, d in zip(x_arr, nd)])


def is_loguniform(frequencies):
    """True when the spacing of ln(f) varies by at most 1 % (std/mean)."""
    steps = np.diff(np.log(frequencies))
    return bool(np.std(steps) / np.mean(steps) <= 0.01)


def get_outlier_thresh(y, iqr_factor=3):
    """IQR outlier threshold (reference utils.py:143-146)."""
    q75, q25 = np.percentile(y, 75), np.percentile(y, 25)
    return q75 + iqr_factor * (q75 - q25)
