"""tests/golden/reference_c1.npz: BASELINE config 1 on its NAMED input.

Config 1 is "BufNMF on bundled Resources/AudioFiles Nicol-LoopE-M.wav, rank 3, fft 1024 / hop 512, 50 iters".  The GPU box
has no /root/reference, so the parity tests ran a synthetic stand-in of the file's length (oracle_np.drum_like); this
script reads the WAV where it lies and stores its samples (data: 453 932 16-bit mono samples at 44.1 kHz -- Jean-Sebastien
Nicol (drums) recorded by Pierre Alexandre Tremblay at Universite de Montreal, 1998; Resources/AudioFiles/-credits.txt:
for demonstration purposes only, which a parity vector is) together with what BOTH oracles make of it at the config's
parameters and NMF seed 42: the frame / bin counts, 48 probe values each of the float bases and activations, their sums.
The reference holds no answer for this job (its outputs start from a random seed; tests/algorithms/public/TestNMF.cpp
asserts repeatability only), so the expected values are the restatements' -- the fixture pins the INPUT and the two
restatements' agreement on it, not the reference's arithmetic.

    python tools/make_reference_c1_fixture.py        (needs /root/reference)
"""
import os
import sys
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
WAV = "/root/reference/Resources/AudioFiles/Nicol-LoopE-M.wav"
WIN, FFT, HOP, K, ITERS, SEED = 1024, 1024, 512, 3, 50, 42


def main():
    import oracle_c
    import oracle_np
    with wave.open(WAV, "rb") as w:
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate()) == (1, 2, 44100)
        pcm = np.frombuffer(w.readframes(w.getnframes()), "<i2")
    assert len(pcm) == 453932
    x = pcm.astype(np.float32) / 32768.0
    o = oracle_c.get("native")
    bc, ac, mag = o.bufnmf_channel(x, WIN, FFT, HOP, K, ITERS, SEED, want_mag=True)
    bn, an, mn, *_ = oracle_np.bufnmf_channel(x, WIN, FFT, HOP, K, ITERS, SEED)
    T, F = mag.shape
    assert (T, F) == (887, 513) and bc.shape == (K, F) and ac.shape == (K, T)
    eb = np.abs(bc - bn).max() / np.abs(bn).max()
    ea = np.abs(ac - an).max() / np.abs(an).max()
    em = np.abs(mag - mn).max() / np.abs(mn).max()
    assert eb < 1e-6 and ea < 1e-6 and em < 1e-12, (eb, ea, em)
    rs = np.random.RandomState(1)
    pb = np.stack([rs.randint(0, K, 48), rs.randint(0, F, 48)], axis=1)
    pa = np.stack([rs.randint(0, K, 48), rs.randint(0, T, 48)], axis=1)
    np.savez_compressed(
        os.path.join(ROOT, "tests", "golden", "reference_c1.npz"),
        pcm16=pcm, sample_rate=np.float64(44100.0), params=np.array([WIN, FFT, HOP, K, ITERS, SEED], dtype=np.int64),
        frames_bins=np.array([T, F], dtype=np.int64),
        probe_bases_idx=pb, probe_acts_idx=pa,
        probe_bases_c=bc[pb[:, 0], pb[:, 1]], probe_acts_c=ac[pa[:, 0], pa[:, 1]],
        probe_bases_np=bn[pb[:, 0], pb[:, 1]], probe_acts_np=an[pa[:, 0], pa[:, 1]],
        sums_c=np.array([bc.astype(np.float64).sum(), ac.astype(np.float64).sum()]),
        sums_np=np.array([bn.astype(np.float64).sum(), an.astype(np.float64).sum()]),
        mag_probe=mag[::97, ::31].copy(),
        source=np.array("flucoma-core Resources/AudioFiles/Nicol-LoopE-M.wav (all 453932 samples)"))
    print("wrote reference_c1.npz: T, F =", T, F, "oracles agree to", eb, ea, em)


if __name__ == "__main__":
    main()
