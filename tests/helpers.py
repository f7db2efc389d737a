import numpy as np


def rel_err(a, b):
    """max |a-b| / max |b| : the normwise relative error the parity bars are stated in."""
    a, b = np.asarray(a), np.asarray(b)
    if not (np.iscomplexobj(a) or np.iscomplexobj(b)):
        a, b = a.astype(np.float64), b.astype(np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def elementwise_rel_err(a, b, floor=1e-6):
    """max over entries of |a - b| / |b|, taken over the entries with |b| > floor * max|b|: north_star's "W/H within 1e-5
    relative" read element by element.  Entries below the floor (components that have decayed towards epsilon over hundreds
    of multiplicative updates carry no digits a float32 output buffer could hold) are covered by the normwise bar only."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    m = np.abs(b) > floor * max(np.abs(b).max(), 1e-300)
    if not m.any():
        return 0.0
    return float((np.abs(a - b)[m] / np.abs(b)[m]).max())


def ulp_histogram(a, b, floor=1e-6):
    """distances in units of the last place of b (float64 spacing), over the entries elementwise_rel_err looks at:
    {"0": n, "1": n, "2-3": n, ..., ">=2^k": n} by powers of two, plus the fraction of entries under the floor"""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    m = np.abs(b) > floor * max(np.abs(b).max(), 1e-300)
    d = np.abs(a - b)[m] / np.spacing(np.abs(b)[m])
    hist = {"0": int((d == 0).sum())}
    lo = 1.0
    while lo <= max(d.max() if d.size else 0.0, 1.0):
        hi = lo * 2.0
        key = str(int(lo)) if lo == 1.0 else f"{int(lo)}-{int(hi) - 1}"
        hist[key] = int(((d >= lo) & (d < hi)).sum()) if lo > 1.0 else int(((d > 0) & (d < 2.0)).sum())
        lo = hi
    return {"entries": int(m.sum()), "below_floor": float(1.0 - m.mean()), "ulps": hist}


# north_star tolerances
TOL_STFT = 1e-12      # f64 vs f64 spectrogram / magnitude
TOL_FACTORS = 1e-5    # W, H relative (BASELINE.json north_star); f64 path is expected << this
TOL_FACTORS_TIGHT = 1e-9
