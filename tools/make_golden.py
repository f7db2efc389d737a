"""Mint the golden fixtures under tests/golden/ (G1..G6).

Run in the build container:  python tools/make_golden.py

The vectors come from the *independent numpy restatement* (oracle/oracle_np.py), never from the
C oracle or the HIP path they are later used to check.  The reference itself cannot be built
here (Eigen / HISSTools / foonathan-memory absent), so these are "parity unpinned" goldens; the
one piece of the path that IS executable verbatim -- libstdc++'s <random> as used by
include/flucoma/algorithms/util/EigenRandom.hpp:73-101 -- is compiled and compared below.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_np as onp  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "golden_v1.npz")

RNG_CPP = r"""
#include <random>
#include <cstdio>
#include <cstdlib>
int main(int argc, char** argv) {
  unsigned long seed = strtoul(argv[1], 0, 10); int n = atoi(argv[2]);
  std::mt19937_64 g{seed}; std::uniform_real_distribution<double> d{0.0, 1.0};
  for (int i = 0; i < n; i++) printf("%.17g\n", d(g));
}
"""


def libstdcxx_uniform(seed, n):
    with tempfile.TemporaryDirectory() as td:
        src, exe = os.path.join(td, "r.cpp"), os.path.join(td, "r")
        open(src, "w").write(RNG_CPP)
        subprocess.run(["g++", "-O1", "-o", exe, src], check=True)
        out = subprocess.run([exe, str(seed), str(n)], check=True, capture_output=True, text=True).stdout
    return np.array([float(x) for x in out.split()])


def test_signal(n, seed):
    """Synthetic fixture in the spirit of tests/test_signals/Signals.cpp.in: sines + impulses."""
    rs = np.random.RandomState(seed)
    t = np.arange(n) / 44100.0
    x = 0.5 * np.sin(2 * np.pi * 440.0 * t) + 0.25 * np.sin(2 * np.pi * 3520.0 * t + 0.3)
    x[n // 3] += 1.0
    x[(2 * n) // 3] -= 0.75
    x += 0.01 * rs.standard_normal(n)
    return x.astype(np.float32)


def main():
    g = {}
    # ---- G1 window tables -----------------------------------------------------------------
    for win in (1024, 2048, 4096):
        w = onp.hann(win)
        g[f"g1_hann{win}_head"] = w[:8]
        g[f"g1_hann{win}_tail"] = w[-8:]
        g[f"g1_hann{win}_sum"] = np.array([w.sum(), (w * w).sum()])
    # ---- G2 STFT ---------------------------------------------------------------------------
    sig = test_signal(8192 + 321, 11)  # ragged length on purpose
    g["g2_signal"] = sig
    for (win, fft, hop) in ((1024, 1024, 512), (2048, 2048, 512), (512, 1024, 256)):
        spec, mag = onp.stft(sig.astype(np.float64), win, fft, hop)
        T = spec.shape[0]
        rows = np.array([0, T // 2, T - 1])
        key = f"g2_{win}_{fft}_{hop}"
        g[key + "_T"] = np.array([T])
        g[key + "_rows"] = rows
        g[key + "_spec"] = spec[rows]
        g[key + "_mag"] = mag[rows]
        g[key + "_magsum"] = np.array([mag.sum(), (mag * mag).sum()])
    # ---- G3 RNG ----------------------------------------------------------------------------
    for seed in (42, 5063):
        ours = onp.rng_uniform01(seed, 4096)
        theirs = libstdcxx_uniform(seed, 4096)
        assert np.array_equal(ours, theirs), "mt19937_64 restatement differs from libstdc++"
        g[f"g3_rng{seed}"] = ours[:16]
        # fill order: W (F=3, K=2): W[f][k] = draw[k*F + f]; H (K=2, T=3): H[k][t] = draw[t*K + k]
        g[f"g3_W{seed}"] = ours[:6].reshape(2, 3).T.copy()
        g[f"g3_H{seed}"] = ours[:6].reshape(3, 2).T.copy()
    # ---- G4 tiny NMF (shape of tests/algorithms/public/TestNMF.cpp:18-27) -------------------
    X = np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9.0]])
    g["g4_X"] = X
    for seed in (42, 5063):
        for iters in (1, 50):
            W1, H1, V1 = onp.nmf_process(X, 2, iters, True, True, seed)
            g[f"g4_s{seed}_i{iters}_W"], g[f"g4_s{seed}_i{iters}_H"], g[f"g4_s{seed}_i{iters}_V"] = W1, H1, V1
    # ---- G5 64 x 33, rank 4 -----------------------------------------------------------------
    rs = np.random.RandomState(5)
    Wt = rs.uniform(0, 1, (4, 33)) ** 3
    Ht = rs.uniform(0, 1, (64, 4)) ** 2
    X5 = Ht @ Wt + 0.01 * rs.uniform(0, 1, (64, 33))
    g["g5_X"] = X5
    W0 = rs.uniform(0.1, 1, (4, 33))
    H0 = rs.uniform(0.1, 1, (64, 4))
    g["g5_W0"], g["g5_H0"] = W0, H0
    for (uw, uh) in ((1, 1), (1, 0), (0, 1), (0, 0)):
        iters = 200 if (uw or uh) else 0
        W1, H1, V1 = onp.nmf_process(X5, 4, iters, bool(uw), bool(uh), 42,
                                     None if uw else W0, None if uh else H0)
        g[f"g5_u{uw}{uh}_W"], g[f"g5_u{uw}{uh}_H"], g[f"g5_u{uw}{uh}_V"] = W1, H1, V1
    W1, H1, V1 = onp.nmf_process(X5, 4, 200, True, True, 42, W0, H0)
    g["g5_seeded_W"], g["g5_seeded_H"], g["g5_seeded_V"] = W1, H1, V1
    # ---- G6 config c1 shape end to end ------------------------------------------------------
    n1 = 453932
    x1 = onp.drum_like(n1)
    g["g6_input_sha256"] = np.frombuffer(hashlib.sha256(x1.tobytes()).digest(), dtype=np.uint8)
    bases, acts, mag, W1, H1, V1 = onp.bufnmf_channel(x1, 1024, 1024, 512, 3, 50, 42)
    assert mag.shape == (887, 513)
    pr = np.random.RandomState(3)
    pb = np.stack([pr.randint(0, 3, 32), pr.randint(0, 513, 32)], axis=1)
    pa = np.stack([pr.randint(0, 3, 32), pr.randint(0, 887, 32)], axis=1)
    g["g6_TF"] = np.array([887, 513])
    g["g6_probe_bases_idx"], g["g6_probe_acts_idx"] = pb, pa
    g["g6_probe_bases"] = bases[pb[:, 0], pb[:, 1]]
    g["g6_probe_acts"] = acts[pa[:, 0], pa[:, 1]]
    g["g6_sums"] = np.array([bases.astype(np.float64).sum(), acts.astype(np.float64).sum(), mag.sum()])
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, **g)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(g), "arrays")


def frames():
    """G7: NMF::processFrame, kept in its own file so that G1..G6 stay byte-identical."""
    import oracle_np as onp
    g = {}
    rs = np.random.RandomState(11)
    K, F, T = 5, 129, 12
    W0 = rs.uniform(0, 1, (K, F)) ** 3
    W0[1, :4] = 0.0
    acts = rs.uniform(0, 2, (T, K)) * (rs.uniform(0, 1, (T, K)) < 0.5)
    X = acts @ W0 + 1e-3 * rs.uniform(0, 1, (T, F))
    X[2, :9] = 0.0
    g["g7_W0"], g["g7_X"] = W0, X
    for seed, iters in ((42, 10), (5063, 10), (42, 0), (7, 100)):
        H = np.empty((T, K)); V = np.empty((T, F))
        for t in range(T):
            H[t], V[t] = onp.nmf_process_frame(X[t], W0, iters, seed)
        g[f"g7_s{seed}_i{iters}_H"], g[f"g7_s{seed}_i{iters}_V"] = H, V
    out = os.path.join(os.path.dirname(OUT), "golden_frames_v1.npz")
    np.savez_compressed(out, **g)
    print("wrote", out, os.path.getsize(out), "bytes,", len(g), "arrays")


if __name__ == "__main__":
    if "--frames" in sys.argv:
        frames()
    else:
        main()
