"""Synthetic 44.1 kHz audio of the benchmark workloads: a sum of exponentially decaying sinusoid
"notes" with random onset / frequency (100 Hz .. 8 kHz) over -40 dB white noise, float32 in [-1, 1] -- non-negative,
non-degenerate, low-rank-ish spectrograms (pure noise gives a flat V and is a poor NMF input).  Buffer b of a corpus
uses seed 1000 + b."""
from __future__ import annotations

import numpy as np


def synth_audio(n: int, seed: int, sr: float = 44100.0, notes: int = 8) -> np.ndarray:
    rs = np.random.RandomState(seed)
    t = np.arange(n, dtype=np.float64) / sr
    x = np.zeros(n)
    dur = n / sr
    for _ in range(notes):
        onset = rs.uniform(0, 0.8 * dur)
        f = np.exp(rs.uniform(np.log(100.0), np.log(8000.0)))
        decay = rs.uniform(2.0, 12.0)
        amp = rs.uniform(0.2, 1.0)
        tt = np.maximum(t - onset, 0.0)
        x += np.where(t >= onset, amp * np.exp(-decay * tt) * np.sin(2 * np.pi * f * tt), 0.0)
    x += 0.01 * rs.standard_normal(n)
    x /= max(1.0, np.abs(x).max())
    return x.astype(np.float32)
