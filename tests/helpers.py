import numpy as np


def rel_err(a, b):
    """max |a-b| / max |b| : the normwise relative error the parity bars are stated in."""
    a, b = np.asarray(a), np.asarray(b)
    if not (np.iscomplexobj(a) or np.iscomplexobj(b)):
        a, b = a.astype(np.float64), b.astype(np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


# north_star tolerances
TOL_STFT = 1e-12      # f64 vs f64 spectrogram / magnitude
TOL_FACTORS = 1e-5    # W, H relative (BASELINE.json north_star); f64 path is expected << this
TOL_FACTORS_TIGHT = 1e-9
