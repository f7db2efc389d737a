"""Independent numpy restatement of flucoma-core's BufNMF hot path.

TEST INFRASTRUCTURE ONLY (see oracle/fluid_oracle.h).  PARITY: the NMF arithmetic is UNPINNED by anything the
reference holds; the STFT -> magnitude -> mel -> DCT chain reproduces the reference's own pre-analysed demo corpus
(Resources/Data/flucoma_corpus_mfcc.json: 299 recomputable slices to the float32 it stores --
tests/test_oracle.py::test_oracles_reproduce_the_references_pre_analysed_corpus).  This file exists so that two independently written restatements (this one: numpy
pocketfft + BLAS matmul; fluid_oracle.c: hand-rolled radix-2 FFT + blocked loops) must agree
to <=1e-12 before either is trusted, and to mint the fixtures under tests/golden/
(tools/make_golden.py).

Reference files followed (relative to /root/reference/include/flucoma/):
  algorithms/public/WindowFuncs.hpp:41-45, algorithms/public/STFT.hpp:90-108,61-66,
  algorithms/util/FFT.hpp:92-108, algorithms/util/EigenRandom.hpp:73-110,
  algorithms/public/NMF.hpp:91-134,144-183, clients/nrt/NMFClient.hpp:233-300.
"""
from __future__ import annotations

import numpy as np

EPS = np.finfo(np.float64).eps  # util/AlgorithmUtils.hpp:19


# --------------------------------------------------------------------------------------
# RNG: std::mt19937_64 + libstdc++ uniform_real_distribution<double>(0, 1)
# --------------------------------------------------------------------------------------
class MT19937_64:
    """std::mt19937_64 (vectorised twist)."""

    NN, MM = 312, 156
    A = np.uint64(0xB5026F5AA96619E9)
    UM = np.uint64(0xFFFFFFFF80000000)
    LM = np.uint64(0x7FFFFFFF)

    def __init__(self, seed: int):
        mt = np.zeros(self.NN, dtype=np.uint64)
        x = int(seed) & 0xFFFFFFFFFFFFFFFF
        mt[0] = x
        for i in range(1, self.NN):
            x = (6364136223846793005 * (x ^ (x >> 62)) + i) & 0xFFFFFFFFFFFFFFFF
            mt[i] = x
        self.mt = mt
        self.idx = self.NN

    def _twist(self):
        mt, NN, MM = self.mt, self.NN, self.MM

        def mix(up, lo, far):
            x = (up & self.UM) | (lo & self.LM)
            return far ^ (x >> np.uint64(1)) ^ np.where(x & np.uint64(1), self.A, np.uint64(0))

        # i in [0, NN-MM): uses old mt[i+1], old mt[i+MM]
        mt[: NN - MM] = mix(mt[: NN - MM], mt[1 : NN - MM + 1], mt[MM:NN])
        # i in [NN-MM, NN-1): uses old mt[i+1], NEW mt[i+MM-NN]
        mt[NN - MM : NN - 1] = mix(mt[NN - MM : NN - 1], mt[NN - MM + 1 : NN], mt[: MM - 1])
        # i = NN-1: uses NEW mt[0], NEW mt[MM-1]
        mt[NN - 1 : NN] = mix(mt[NN - 1 : NN], mt[0:1], mt[MM - 1 : MM])
        self.idx = 0

    def raw(self, count: int) -> np.ndarray:
        out = np.empty(count, dtype=np.uint64)
        done = 0
        while done < count:
            if self.idx >= self.NN:
                self._twist()
            take = min(count - done, self.NN - self.idx)
            out[done : done + take] = self.mt[self.idx : self.idx + take]
            self.idx += take
            done += take
        x = out
        x = x ^ ((x >> np.uint64(29)) & np.uint64(0x5555555555555555))
        x = x ^ ((x << np.uint64(17)) & np.uint64(0x71D67FFFEDA60000))
        x = x ^ ((x << np.uint64(37)) & np.uint64(0xFFF7EEE000000000))
        x = x ^ (x >> np.uint64(43))
        return x


def rng_uniform01(seed: int, count: int) -> np.ndarray:
    """libstdc++ generate_canonical<double,53> over mt19937_64: double(u64) / 2**64."""
    u = MT19937_64(seed).raw(count)
    r = u.astype(np.float64) / 18446744073709551616.0  # u64 -> f64 is round-to-nearest
    r[r >= 1.0] = np.nextafter(1.0, 0.0)
    return r


# --------------------------------------------------------------------------------------
# STFT
# --------------------------------------------------------------------------------------
def hann(win: int) -> np.ndarray:
    i = np.arange(win, dtype=np.float64)
    return 0.5 - 0.5 * np.cos((np.pi * 2 * i) / win)


def stft_num_frames(n: int, win: int, hop: int) -> int:
    return (n + hop) // hop


def stft(audio: np.ndarray, win: int, fft: int, hop: int):
    """Returns (spec complex128 [T,F], mag float64 [T,F])."""
    audio = np.asarray(audio, dtype=np.float64)
    n = audio.shape[0]
    padded = np.zeros(n + win + hop)
    padded[win // 2 : win // 2 + n] = audio
    T = (padded.shape[0] - win) // hop
    w = hann(win)
    idx = np.arange(T)[:, None] * hop + np.arange(win)[None, :]
    frames = padded[idx] * w[None, :]
    spec = np.fft.rfft(frames, n=fft, axis=1)  # zero-pads the tail when win < fft
    spec[:, 0] = spec[:, 0].real
    spec[:, -1] = spec[:, -1].real
    return spec, np.abs(spec)


# --------------------------------------------------------------------------------------
# NMF
# --------------------------------------------------------------------------------------
def nmf_process(X, K, iters, updateW=True, updateH=True, seed=42, W0=None, H0=None,
                progress=None):
    """X: [T,F].  Returns W1 [K,F], H1 [T,K], V1 [T,F] (alg/NMF.hpp:91-134)."""
    X = np.asarray(X, dtype=np.float64)
    T, F = X.shape
    if W0 is None:
        W = rng_uniform01(seed, F * K).reshape(K, F).T.copy()  # column-major F x K fill
    else:
        W = np.asarray(W0, dtype=np.float64).T.copy()
    if H0 is None:
        H = rng_uniform01(seed, K * T).reshape(T, K).T.copy()  # column-major K x T fill
    else:
        H = np.asarray(H0, dtype=np.float64).T.copy()
    V = X.T.copy()
    H = np.maximum(H, EPS)
    W = np.maximum(W, EPS)
    W = W / np.sqrt((W * W).sum(axis=0, keepdims=True))
    H = H / np.sqrt((H * H).sum(axis=1, keepdims=True))
    cancelled = False
    for it in range(iters):
        if updateW:
            V1 = np.maximum(W @ H, EPS)
            wnum = (V / V1) @ H.T
            wden = H.sum(axis=1)[None, :]
            W = W * wnum / np.maximum(wden, EPS)
            if W.max() > EPS:
                W = W / np.sqrt((W * W).sum(axis=0, keepdims=True))
        V2 = np.maximum(W @ H, EPS)
        if updateH:
            hnum = W.T @ (V / V2)
            hden = W.sum(axis=0)[:, None]
            H = H * hnum / np.maximum(hden, EPS)
        if progress is not None and not progress(it + 1):
            cancelled = True
            break
    Vout = V if cancelled else W @ H
    return W.T.copy(), H.T.copy(), Vout.T.copy()


def nmf_process_frame(x, W0, iters, seed):
    """alg/NMF.hpp:45-89: activations h [K] of the dictionary W0 [K,F] in one magnitude frame x [F], and the
    estimate W^T h [F]."""
    W = np.maximum(np.asarray(W0, dtype=np.float64), EPS)
    K = W.shape[0]
    h = np.maximum(rng_uniform01(seed, K), EPS)
    v0 = np.maximum(np.asarray(x, dtype=np.float64), EPS)
    W = W / np.sqrt((W * W).sum(axis=1, keepdims=True))
    den = np.maximum(W.sum(axis=1), EPS)
    for _ in range(iters):
        v1 = np.maximum(W.T @ h, EPS)
        h = h * (W @ (v0 / v1)) / den
    return h, W.T @ h


def nndsvd_rank(s, min_rank, max_rank, amount):
    """alg/NNDSVD.hpp:47-58: smallest k whose leading singular values cover `amount` of their sum, clamped."""
    if amount == 0:
        k = min_rank
    else:
        k, current, total = 0, 0.0, float(np.sum(s))
        while current / total < amount:
            current += float(s[k])
            k += 1
    return int(min(max(k, min_rank), max_rank))


def nndsvd_from_svd(U, s, VT, X, W_rows, min_rank, max_rank, amount, method, seed):
    """alg/NNDSVD.hpp:44-129 after the SVD: U [F,r], s [r], VT [r,T] of X^T (F x T).  Returns W [W_rows,F],
    H [T,W_rows] (rows / columns beyond k stay zero like the reference's zero-initialised outputs) and k.
    Methods 1..3 depend on the sign convention of the SVD (and carry the reference's `yNNorm = xN.norm()`,
    :85); method 0 (|U|, |S V^T|) does not."""
    F, T = U.shape[0], VT.shape[1]
    k = nndsvd_rank(s, min_rank, max_rank, amount)
    WT = np.zeros((F, W_rows))
    HT = np.zeros((W_rows, T))
    if method == 0:
        WT[:, :k] = np.abs(U[:, :k])
        HT[:k, :] = np.abs(s[:k, None] * VT[:k, :])
    else:
        WT[:, 0] = np.abs(U[:, 0])
        HT[0, :] = np.sqrt(s[0]) * np.abs(VT[0, :])
        for j in range(1, k):
            x, y = U[:, j], VT[j, :]
            xP, yP = np.maximum(x, 0.0), np.maximum(y, 0.0)
            xN, yN = np.abs(np.minimum(x, 0.0)), np.abs(np.minimum(y, 0.0))
            xPn, yPn, xNn = np.sqrt((xP * xP).sum()), np.sqrt((yP * yP).sum()), np.sqrt((xN * xN).sum())
            yNn = xNn                                       # :85 as written in the reference
            mP, mN = xPn * yPn, xNn * yNn
            if mP > mN:
                u, v, sigma = xP / xPn, yP / yPn, mP
            else:
                u, v, sigma = xN / xNn, yN / yNn, mN
            WT[:, j] = u
            HT[j, :] = np.sqrt(s[j] * sigma) * v
        mean = float(np.mean(X))
        if method == 1:
            # :107-116: EigenRandom(..., Range{eps, mean*0.001}) is a lazy NullaryExpr inside select(): the generator
            # is only called where the condition holds, in the assignment's column-major traversal -- the n-th
            # draw lands on the n-th sub-epsilon coefficient; (max - min) * u + min like libstdc++
            lo, hi = EPS, mean * 0.001
            for M_ in (WT, HT):
                flat = M_.T.reshape(-1)                      # column-major view of M_
                idx = np.flatnonzero(flat < EPS)
                draws = (hi - lo) * rng_uniform01(seed, idx.size) + lo
                colmajor = M_.T.copy().reshape(-1)
                colmajor[idx] = draws
                M_[...] = colmajor.reshape(M_.shape[1], M_.shape[0]).T
        elif method == 2:
            WT = np.where(WT < EPS, mean, WT)
            HT = np.where(HT < EPS, mean, HT)
    return WT.T.copy(), HT.T.copy(), k


def nndsvd(X, W_rows, min_rank=0, max_rank=200, amount=0.8, method=0, seed=-1):
    """alg/NNDSVD.hpp:30-132 with LAPACK's SVD in place of Eigen's BDCSVD.  X: [T,F] magnitudes."""
    X = np.asarray(X, dtype=np.float64)
    U, s, VT = np.linalg.svd(X.T, full_matrices=False)
    return nndsvd_from_svd(U, s, VT, X, W_rows, min_rank, max_rank, amount, method, seed) + (U, s, VT)


def bufnmf_writeback(W1, H1):
    """nrt/NMFClient.hpp:277-300: bases [K,F] f32; activations [K,T] f32 (float multiply)."""
    bases = W1.astype(np.float32)
    scale = 1.0 / H1.max()
    acts = H1.T.astype(np.float32) * np.float32(scale)
    return bases, acts.astype(np.float32)


def bufnmf_channel(audio_f32, win, fft, hop, K, iters, seed):
    audio = np.asarray(audio_f32, dtype=np.float32).astype(np.float64)
    _, mag = stft(audio, win, fft, hop)
    W1, H1, V1 = nmf_process(mag, K, iters, True, True, seed)
    bases, acts = bufnmf_writeback(W1, H1)
    return bases, acts, mag, W1, H1, V1


# --------------------------------------------------------------------------------------
# resynthesis
# --------------------------------------------------------------------------------------
def resynth_component(spec, W1, H1, V1, k, win, fft, hop, n):
    T, F = spec.shape
    est = np.outer(H1[:, k], W1[k, :])
    mask = np.minimum(est * (1.0 / np.maximum(V1, EPS)), 1.0)
    Y = spec * mask
    frames = np.fft.irfft(Y, n=fft, axis=1)[:, :win]  # = unnormalised inverse * (1/fft)
    w = hann(win)
    outsz = win + (T - 1) * hop + win + hop
    acc = np.zeros(outsz)
    nrm = np.zeros(outsz)
    for t in range(T):
        acc[t * hop : t * hop + win] += frames[t] * w
        nrm[t * hop : t * hop + win] += w * w
    out = acc / np.maximum(nrm, EPS)
    return out[win // 2 : win // 2 + n]


# --------------------------------------------------------------------------------------
# synthetic audio (SURVEY 8 d): decaying sinusoid "notes" + -40 dB noise, float32
# --------------------------------------------------------------------------------------
# the workload generator is product-side data plumbing (bench.py, tools/), not a checker: it lives in the package and
# is re-exported here so the tests and the fixtures' generator script keep one definition
import os as _os
import sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "flucoma-core_amd"))
from synth import synth_audio  # noqa: E402,F401


def drum_like(n, seed=7):
    """c1-shaped stand-in for Resources/AudioFiles/Nicol-LoopE-M.wav (which is not redistributable):
    three kinds of decaying hits (low thump, mid noise burst, high tick) on a loop grid."""
    rs = np.random.RandomState(seed)
    sr = 44100.0
    x = np.zeros(n)
    step = int(sr * 0.125)
    pos = 0
    i = 0
    while pos < n:
        kind = [0, 2, 1, 2, 0, 0, 1, 2][i % 8]
        L = min(n - pos, int(sr * 0.25))
        tt = np.arange(L) / sr
        if kind == 0:
            hit = np.exp(-18 * tt) * np.sin(2 * np.pi * (55 + 60 * np.exp(-30 * tt)) * tt)
        elif kind == 1:
            hit = 0.6 * np.exp(-25 * tt) * rs.standard_normal(L) + 0.3 * np.exp(-20 * tt) * np.sin(2 * np.pi * 190 * tt)
        else:
            hit = 0.35 * np.exp(-60 * tt) * rs.standard_normal(L)
        x[pos:pos + L] += hit
        pos += step
        i += 1
    x += 0.002 * rs.standard_normal(n)
    x /= np.abs(x).max() * 1.05
    # 16-bit quantisation like the WAV it stands in for
    return (np.round(x * 32767.0) / 32768.0).astype(np.float32)


# --------------------------------------------------------------------------------------
# MelBands / DCT / MFCC: alg/MelBands.hpp:36-97, alg/DCT.hpp:36-75,
# rt/MFCCClient.hpp:86-131, rt/MelBandsClient.hpp:77-119, StreamingControl framing
# (cc/FluidNRTClientWrapper.hpp:551-660)
# --------------------------------------------------------------------------------------
def feature_padding(win: int, hop: int, padding_mode: int) -> int:
    """FFTParams::padding, cc/ParameterTypes.hpp:315-323: None / Default / Full."""
    return [0, win >> 1, win - hop][padding_mode]


def feature_frames(n: int, win: int, hop: int, padding_mode: int = 1):
    """(T, first-sample offset of frame 0) of the buffered feature clients (StreamingControl,
    cc/FluidNRTClientWrapper.hpp:551-660): the audio sits userPadding = FFTParams::padding samples into a padded signal
    of n + latency + 2 userPadding samples (latency = win; rounded up to whole hops in Full mode, :572-574),
    nAnalysisFrames = 1 + (paddedLength - win) // hop, analysis frame j sees padded[(j + 1) hop - win, (j + 1) hop)
    ... of the client's FluidSource, whose delay puts kept frame k (the first win // hop are dropped, :643-656) at audio
    sample latencyHops hop - win - userPadding + k hop (streaming_control_frame_starts RUNS the model and agrees)."""
    pad = feature_padding(win, hop, padding_mode)
    padded = n + win + 2 * pad
    if padding_mode == 2:
        padded = -(-padded // hop) * hop
    n_analysis = 1 + (padded - win) // hop
    latency_hops = win // hop
    T = n_analysis - latency_hops
    start0 = latency_hops * hop - win - pad
    return T, start0


def framed_magnitude(audio, win, fft, hop, padding_mode=1):
    audio = np.asarray(audio, dtype=np.float64)
    n = audio.shape[0]
    T, start0 = feature_frames(n, win, hop, padding_mode)
    w = hann(win)
    idx = start0 + np.arange(T)[:, None] * hop + np.arange(win)[None, :]
    ok = (idx >= 0) & (idx < n)
    frames = np.where(ok, audio[np.clip(idx, 0, n - 1)], 0.0) * w[None, :]
    return np.abs(np.fft.rfft(frames, n=fft, axis=1))


def mel_filters(lo, hi, n_bands, n_bins, sr):
    mel = lambda x: 1127.01048 * np.log(x / 700.0 + 1.0)
    centres = 700.0 * (np.exp(np.linspace(mel(lo), mel(hi), n_bands + 2) / 1127.01048) - 1.0)
    hz = np.linspace(0.0, sr / 2.0, n_bins)
    d = np.abs(centres[:-1] - centres[1:])
    lower = (hz[None, :] - centres[:n_bands, None]) / d[:n_bands, None]
    upper = (centres[2:, None] - hz[None, :]) / d[1:, None]
    return np.maximum(0.0, np.minimum(lower, upper))


def dct_table(n_in, n_out):
    i = np.arange(n_out)[:, None]
    j = np.linspace(0.5, n_in - 0.5, n_in)[None, :]
    scale = np.where(i == 0, 1.0 / np.sqrt(n_in), np.sqrt(2.0 / n_in))
    return np.cos((np.pi / n_in) * i * j) * scale


def melbands(mag, filt, win, mag_norm, use_power, log_output):
    F = mag.shape[1]
    frame = mag * (1.0 / (win / 4.0)) if mag_norm else mag
    energy = frame.sum(axis=1) * (1.0 / (2.0 * (2 * (F - 1)) / win))
    if use_power:
        frame = frame * frame
    out = frame @ filt.T
    if mag_norm:
        out = out * energy[:, None] / np.maximum(EPS, out.sum(axis=1))[:, None]
    if log_output:
        out = 20.0 * np.log10(np.maximum(out, EPS))
    return out


def bufmelbands_channel(audio_f32, win, fft, hop, n_bands=40, lo=20.0, hi=20000.0, sr=44100.0,
                        normalize=True, scale_db=False, padding_mode=1):
    mag = framed_magnitude(np.asarray(audio_f32, dtype=np.float32).astype(np.float64), win, fft, hop, padding_mode)
    filt = mel_filters(lo, hi, n_bands, fft // 2 + 1, sr)
    return melbands(mag, filt, win, normalize, False, scale_db).T.astype(np.float32)


def bufmfcc_channel(audio_f32, win, fft, hop, n_bands=40, n_coefs=13, start_coeff=0, lo=20.0, hi=20000.0,
                    sr=44100.0, padding_mode=1):
    mag = framed_magnitude(np.asarray(audio_f32, dtype=np.float32).astype(np.float64), win, fft, hop, padding_mode)
    filt = mel_filters(lo, hi, n_bands, fft // 2 + 1, sr)
    bands = melbands(mag, filt, win, False, False, True)
    n_out = min(n_coefs + start_coeff, n_bands)
    coefs = bands @ dct_table(n_bands, n_out).T
    out = np.zeros((mag.shape[0], n_coefs))
    take = max(0, min(n_coefs, n_out - start_coeff))
    out[:, :take] = coefs[:, start_coeff:start_coeff + take]
    return out.T.astype(np.float32)


# --------------------------------------------------------------------------------------
# BufSTFT: nrt/BufSTFTClient.hpp:81-276
# --------------------------------------------------------------------------------------
def bufstft_padding(win, hop, mode):
    return [0, win >> 1, win - hop][mode]


def bufstft_forward(audio_f32, win, fft, hop, padding_mode=1):
    x = np.asarray(audio_f32, dtype=np.float32).astype(np.float64)
    n = x.shape[0]
    pad = bufstft_padding(win, hop, padding_mode)
    padded_len = n + 2 * pad
    if padding_mode == 2:
        padded_len = -(-padded_len // hop) * hop
    padded = np.zeros(padded_len)
    padded[pad:pad + n] = x
    T = 1 + (padded_len - win) // hop
    idx = np.arange(T)[:, None] * hop + np.arange(win)[None, :]
    spec = np.fft.rfft(padded[idx] * hann(win)[None, :], n=fft, axis=1)
    spec[:, 0] = spec[:, 0].real
    spec[:, -1] = spec[:, -1].real
    return np.abs(spec).T.astype(np.float32), np.angle(spec).T.astype(np.float32)


def bufstft_inverse(mag, phase, win, fft, hop, padding_mode=1):
    mag = np.asarray(mag, dtype=np.float32)
    phase = np.asarray(phase, dtype=np.float32)
    F, T = mag.shape
    pad = bufstft_padding(win, hop, padding_mode)
    # nrt/BufSTFTClient.hpp:248-250: std::polar on the float samples = single-precision m cos p, m sin p, widened after
    spec = ((mag * np.cos(phase)).astype(np.float64) + 1j * (mag * np.sin(phase)).astype(np.float64)).T
    frames = np.fft.irfft(spec, n=fft, axis=1)[:, :win] * hann(win)[None, :]
    size = (T - 1) * hop + win
    acc, nrm = np.zeros(size), np.zeros(size)
    w2 = hann(win) ** 2
    for t in range(T):
        acc[t * hop:t * hop + win] += frames[t]
        nrm[t * hop:t * hop + win] += w2
    return (acc / np.maximum(nrm, EPS))[pad:]


# --------------------------------------------------------------------------------------
# FluidSource / FluidSink / BufferedProcess (cc/FluidSource.hpp:20-175, cc/FluidSink.hpp, cc/BufferedProcess.hpp:35-112):
# the ring buffers every buffered real-time client frames its input with.  Restated literally so that the
# reference's own known-answer tests of them (tests/clients/common/TestFluidSource.cpp:17-59,
# TestBufferedProcess.cpp:20-70) can be replayed, and so that the closed-form frame bookkeeping of
# feature_frames() above is DERIVED from the ring-buffer behaviour instead of asserted.
# --------------------------------------------------------------------------------------
class FluidSourceModel:
    """cc/FluidSource.hpp: single channel; buffer of size + hostSize samples; push appends a host block at the
    write counter (:117-127, counter advance in copyIn :159-170), pull(frame, frameTime) reads the `blocksize` samples
    that END hostSize - frameTime samples behind the write counter (:68-90)."""

    def __init__(self, size, host_size, dtype=np.float64):
        self.size, self.host = size, host_size
        self.buf = np.zeros(size + host_size, dtype=dtype)
        self.counter = 0

    def buffer_size(self):
        return self.size + self.host

    def push(self, block):
        block = np.asarray(block)
        bs, B = block.shape[0], self.buffer_size()
        assert bs <= B
        off = self.counter
        size = B - off if off + bs > B else bs
        if size:                               # copyIn (:159-170): the counter moves only when something was copied,
            self.buf[off:off + size] = block[:size]     # to offset + size -- it may rest AT bufferSize, never wraps itself
            self.counter = off + size
        if bs - size:
            self.buf[:bs - size] = block[size:]
            self.counter = bs - size

    def pull(self, blocksize, frame_time):
        B = self.buffer_size()
        offset = self.host - frame_time
        if offset > B:
            return np.zeros(blocksize, dtype=self.buf.dtype)
        offset += blocksize
        offset = self.counter - offset if offset <= self.counter else self.counter + B - offset
        size = B - offset if offset + blocksize > B else blocksize
        return np.concatenate([self.buf[offset:offset + size], self.buf[:blocksize - size]])


class FluidSinkModel:
    """cc/FluidSink.hpp: overlap-add ring of size + hostSize samples; push(frame, frameTime) adds a frame starting
    frameTime samples after the read counter, pull hands out the next host block and zeroes it behind itself."""

    def __init__(self, size, host_size):
        self.size, self.host = size, host_size
        self.buf = np.zeros(size + host_size)
        self.counter = 0

    def push(self, frame, frame_time):
        B = self.buf.shape[0]
        bs = frame.shape[0]
        assert bs <= B
        off = frame_time
        if off + bs > B:                       # cc/FluidSink.hpp:52: a frame that would lap the ring is dropped
            return
        off += self.counter
        off = off if off < B else off - B
        size = B - off if off + bs > B else bs
        self.buf[off:off + size] += frame[:size]
        self.buf[:bs - size] += frame[size:]

    def pull(self, blocksize):
        B = self.buf.shape[0]
        off = self.counter
        size = B - off if off + blocksize > B else blocksize
        out = np.concatenate([self.buf[off:off + size], self.buf[:blocksize - size]])
        if size:                               # outAndZero (:141-153)
            self.buf[off:off + size] = 0
            self.counter = off + size
        if blocksize - size:
            self.buf[:blocksize - size] = 0
            self.counter = blocksize - size
        return out


def streaming_control_frame_starts(n, win, hop, padding_mode=1):
    """First audio sample index of every frame the offline-wrapped feature clients keep, found by RUNNING the model:
    StreamingControl (cc/FluidNRTClientWrapper.hpp:551-660) pads the input by win/2 in front (userPadding.first for
    the default padding of the analysis clients), feeds it to the client in host blocks of `hop` samples
    (c.hostVectorSize(controlRate), :575), the client's BufferedProcess pulls a win-sample frame per hop from its
    FluidSource (cc/BufferedProcess.hpp:78-95), and the first latency/hop output columns are dropped (:643-656).
    The padded signal here carries sample INDICES (audio sample i -> value i + 1, padding -> 0), so the pulled frames
    say where they came from.  Returns (T, [start sample of kept frame k])."""
    pad = feature_padding(win, hop, padding_mode)  # FFTParams::padding (cc/ParameterTypes.hpp:315-323)
    latency = win                                  # analysis clients: latency() = winSize
    padded_len = n + latency + 2 * pad             # :564-569: totalPadding = latency + 2 userPadding.first; audio at [pad, pad + n)
    if padding_mode == 2:                          # :572-574
        padded_len = -(-padded_len // hop) * hop
    n_analysis = 1 + (padded_len - win) // hop
    padded = np.zeros(padded_len + hop, dtype=np.int64)
    padded[pad:pad + n] = np.arange(1, n + 1)
    src = FluidSourceModel(win, hop, dtype=np.int64)
    frame_time = 0
    frames = []
    for j in range(n_analysis):
        src.push(padded[j * hop:(j + 1) * hop])
        while frame_time < hop:                    # BufferedProcess::processInput
            frames.append(src.pull(win, frame_time))
            frame_time += hop
        frame_time -= hop
    latency_hops = latency // hop
    kept = frames[latency_hops:]
    starts = []
    for f in kept:
        nz = np.flatnonzero(f)
        # a frame lying wholly in the padding carries no index; callers only compare frames that touch the audio
        starts.append(int(f[nz[0]] - 1 - nz[0]) if nz.size else None)
    return n_analysis - latency_hops, starts


# --------------------------------------------------------------------------------------
# The users of NMF::processFrame: the real-time clients NMFMatch and NMFFilter (clients/rt/NMFMatchClient.hpp:76-118,
# NMFFilterClient.hpp:69-118), restated LITERALLY -- host vector in, host vector out, FluidSource / FluidSink rings,
# one processFrame per hop -- and driven offline the way the reference's own wrapper templates drive a real-time client
# (cc/FluidNRTClientWrapper.hpp: StreamingControl :551-660 for control-rate outputs, Streaming :466-547 for audio
# outputs).  The closed forms below them are what the HIP entry points (fluhip_nmfmatch_f32 / fluhip_nmffilter_f32)
# compute in one batch; tests/test_oracle.py holds the two against each other.
# --------------------------------------------------------------------------------------
def _stft_frame(frame, w, fft):
    """STFT::processFrame (alg/STFT.hpp:110-121): window, zero-pad to the transform size, real FFT (util/FFT.hpp:92-108)"""
    X = np.fft.rfft(frame * w, n=fft)
    X[0] = X[0].real
    X[-1] = X[-1].real
    return X


def _process_frame_w(W0):
    """the dictionary as processFrame leaves it (NMF.hpp:59, 64-65): clamped, every component divided by its L2 norm"""
    W = np.maximum(np.asarray(W0, dtype=np.float64), EPS)
    return W / np.sqrt((W * W).sum(axis=1, keepdims=True))


class NMFMatchClientModel:
    """rt/NMFMatchClient.hpp.  process(block) -> the control output of this call: the activations AS THEY STAND WHEN THE
    CALL STARTS (:104, written before processInput runs), zeros beyond the rank; then every frame the block completes
    updates them by processFrame with TEN iterations (:113-116: the literal 10, the `iterations` parameter is not read)."""

    def __init__(self, bases, max_rank, win, fft, hop, seed, host_size):
        self.bases = np.asarray(bases, dtype=np.float32)         # [K, F] as the filter buffer's channels hold them
        self.max_rank, self.win, self.fft, self.hop, self.seed = max_rank, win, fft, hop, seed
        self.w = hann(win)
        self.host = host_size
        self.reset()
        self.act = np.zeros(max_rank)                            # mActivations(maxRank): value-initialised

    def reset(self):                                             # :74 mSTFTProcessor.reset(): the ring and the frame clock
        self.src = FluidSourceModel(self.win, self.host)
        self.frame_time = 0

    def process(self, block):
        F = self.fft // 2 + 1
        out = np.zeros(self.max_rank)
        if self.bases.shape[1] != F:                             # :91 wrong frame size: nothing happens
            return out
        rank = min(self.bases.shape[0], self.max_rank)
        out[:rank] = self.act[:rank]                             # :104-105
        self.src.push(np.asarray(block, dtype=np.float64))
        while self.frame_time < self.host:                       # BufferedProcess::processInput
            X = _stft_frame(self.src.pull(self.win, self.frame_time), self.w, self.fft)
            h, _ = nmf_process_frame(np.abs(X), self.bases[:rank].astype(np.float64), 10, self.seed)
            self.act[:rank] = h
            self.frame_time += self.hop
        self.frame_time -= self.host
        return out


def nmfmatch_streaming_control(audio_f32, bases, win, fft, hop, seed, padding_mode=1, max_rank=None):
    """NMFMatch behind StreamingControl, literally: padded copy, one client call per hop (host vector = hop, :580), column j of
    the output = what call j returned, the first latency / hop columns dropped.  -> float32 [rank, keepHops]."""
    audio = np.asarray(audio_f32, dtype=np.float64)
    n = audio.shape[0]
    K = np.asarray(bases).shape[0]
    max_rank = K if max_rank is None else max_rank
    pad = feature_padding(win, hop, padding_mode)
    padded_len = n + win + 2 * pad
    if padding_mode == 2:
        padded_len = -(-padded_len // hop) * hop
    n_analysis = 1 + (padded_len - win) // hop
    padded = np.zeros(padded_len + hop)
    padded[pad:pad + n] = audio
    client = NMFMatchClientModel(bases, max_rank, win, fft, hop, seed, hop)
    cols = [client.process(padded[j * hop:(j + 1) * hop]) for j in range(n_analysis)]
    rank = min(K, max_rank)
    lat = win // hop
    return np.stack(cols[lat:], axis=1)[:rank].astype(np.float32) if n_analysis > lat else np.zeros((rank, 0), np.float32)


def nmfmatch_channel(audio_f32, bases, win, fft, hop, seed, padding_mode=1, max_rank=None):
    """The same in closed form: kept column k is processFrame (10 iterations) of the frame at audio sample
    (k + latencyHops - 1) hop - win - userPadding -- ONE HOP BEHIND the frame the analysis clients (BufMFCC) put in that
    column, because the output is written before the call's frame is processed; a column with no frame behind it yet
    (k + latencyHops = 0) holds the initial zeros."""
    audio = np.asarray(audio_f32, dtype=np.float64)
    n = audio.shape[0]
    bases = np.asarray(bases, dtype=np.float32)
    K = bases.shape[0]
    rank = min(K, K if max_rank is None else max_rank)
    T, _ = feature_frames(n, win, hop, padding_mode)
    pad = feature_padding(win, hop, padding_mode)
    lat = win // hop
    w = hann(win)
    out = np.zeros((rank, max(T, 0)), dtype=np.float32)
    for k in range(T):
        f = k + lat - 1
        if f < 0:
            continue
        idx = f * hop - win - pad + np.arange(win)
        ok = (idx >= 0) & (idx < n)
        frame = np.where(ok, audio[np.clip(idx, 0, n - 1)], 0.0)
        h, _ = nmf_process_frame(np.abs(_stft_frame(frame, w, fft)), bases[:rank].astype(np.float64), 10, seed)
        out[:, k] = h
    return out


class NMFFilterClientModel:
    """rt/NMFFilterClient.hpp.  process(block) -> [rank, len(block)]: per completed frame, processFrame (`iterations`),
    the estimate W^T h as the ratio mask's denominator (:104), component i's rank-one estimate through the mask
    (alg/RatioMask.hpp:39-56, exponent 1), inverse frame (ISTFT::processFrame: inverse FFT, 1 / fft, window), overlap-add;
    an extra channel overlap-adds window^2 and normalises the pulled block (BufferedProcess.hpp:219-239: x /= g where
    x != 0, g > 0)."""

    def __init__(self, bases, win, fft, hop, iters, seed, host_size):
        self.bases = np.asarray(bases, dtype=np.float32)
        self.win, self.fft, self.hop, self.iters, self.seed, self.host = win, fft, hop, iters, seed, host_size
        self.w = hann(win)
        self.reset()

    def reset(self):
        K = self.bases.shape[0]
        self.src = FluidSourceModel(self.win, self.host)
        self.sinks = [FluidSinkModel(self.win, self.host) for _ in range(K + 1)]
        self.frame_time = 0

    def process(self, block):
        K, F = self.bases.shape
        nb = len(block)
        if F != self.fft // 2 + 1:
            return np.zeros((K, nb))
        self.src.push(np.asarray(block, dtype=np.float64))
        while self.frame_time < self.host:                       # BufferedProcess::process
            X = _stft_frame(self.src.pull(self.win, self.frame_time), self.w, self.fft)
            h, vhat = nmf_process_frame(np.abs(X), self.bases.astype(np.float64), self.iters, self.seed)
            Wn = _process_frame_w(self.bases)                    # processFrame normalised tmpFilt in place (:50-65)
            mult = 1.0 / np.maximum(vhat, EPS)                   # RatioMask::init
            for i in range(K):
                Y = X * np.minimum(1.0, Wn[i] * h[i] * mult)     # NMF::estimate + RatioMask::process
                y = np.fft.irfft(Y, n=self.fft)[:self.win] * self.w
                self.sinks[i].push(y, self.frame_time)
            self.sinks[K].push(self.w * self.w, self.frame_time)
            self.frame_time += self.hop
        self.frame_time -= self.host
        g = self.sinks[K].pull(nb)
        out = np.empty((K, nb))
        for i in range(K):
            x = self.sinks[i].pull(nb)
            out[i] = np.where(x != 0, x / np.where(g > 0, g, 1.0), x)
        return out


def nmffilter_streaming(audio_f32, bases, win, fft, hop, iters, seed, host=64):
    """NMFFilter behind Streaming (cc/FluidNRTClientWrapper.hpp:466-547), literally: host vectors of 64 samples
    (NRTClientWrapper::VectorSize, :195), the input followed by zeros up to a whole number of vectors covering
    nFrames + latency, the first `latency` = win output samples dropped.  -> float32 [K, n]."""
    audio = np.asarray(audio_f32, dtype=np.float64)
    n = audio.shape[0]
    K = np.asarray(bases).shape[0]
    n_hops = -(-(n + win) // host)
    inp = np.zeros(host * n_hops)
    inp[:n] = audio
    client = NMFFilterClientModel(bases, win, fft, hop, iters, seed, host)
    out = np.concatenate([client.process(inp[j * host:(j + 1) * host]) for j in range(n_hops)], axis=1)
    return out[:, win:win + n].astype(np.float32)


def nmffilter_channel(audio_f32, bases, win, fft, hop, iters, seed):
    """The same in closed form: frames m = 1, 2, ... at audio samples m hop - win (the ring's delay of one window), each
    masked per component and overlap-added where it came from; divided by the sum of window^2 over the frames covering a
    sample (which is never zero where the numerator is not).  -> float32 [K, n]."""
    audio = np.asarray(audio_f32, dtype=np.float64)
    n = audio.shape[0]
    bases = np.asarray(bases, dtype=np.float32)
    K = bases.shape[0]
    w = hann(win)
    Wn = _process_frame_w(bases)
    acc = np.zeros((K, n + 2 * win))
    nrm = np.zeros(n + 2 * win)
    m = 1
    while m * hop - win < n:
        s = m * hop - win
        idx = s + np.arange(win)
        ok = (idx >= 0) & (idx < n)
        frame = np.where(ok, audio[np.clip(idx, 0, n - 1)], 0.0)
        X = _stft_frame(frame, w, fft)
        h, vhat = nmf_process_frame(np.abs(X), bases.astype(np.float64), iters, seed)
        mult = 1.0 / np.maximum(vhat, EPS)
        for i in range(K):
            y = np.fft.irfft(X * np.minimum(1.0, Wn[i] * h[i] * mult), n=fft)[:win] * w
            acc[i, s + win:s + 2 * win] += y
        nrm[s + win:s + 2 * win] += w * w
        m += 1
    out = acc / np.maximum(nrm, EPS)[None, :]
    return out[:, win:win + n].astype(np.float32)
