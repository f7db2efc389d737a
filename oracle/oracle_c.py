"""Build + ctypes binding of oracle/fluid_oracle.c.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product (flucoma-core_amd/) never imports this.

The shared object is compiled on the machine that uses it (gcc is in the image on both the
build container and the GPU box) and cached per (flags, CPU-flag hash) so a `-march=native`
build made on one host is never loaded on a different CPU.
"""
from __future__ import annotations

import ctypes
import hashlib
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")

FLAVOURS = {
    # the reference ships Linux x86 builds with -msse4 (script/flucoma_simdcmd.cmake:20-22)
    "sse4": ["-O3", "-msse4.2"],
    # README.md:56-58 suggests -DFLUID_ARCH=-mnative for local builds
    "native": ["-O3", "-march=native"],
}


def _cpu_tag() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return hashlib.sha1(line.encode()).hexdigest()[:10]
    except OSError:
        pass
    return "unknown"


def build(flavour: str = "native", force: bool = False) -> str:
    flags = FLAVOURS[flavour]
    tag = _cpu_tag() if flavour == "native" else "generic"
    os.makedirs(_BUILD, exist_ok=True)
    out = os.path.join(_BUILD, f"libfluid_oracle_{flavour}_{tag}.so")
    src = os.path.join(_HERE, "fluid_oracle.c")
    hdr = os.path.join(_HERE, "fluid_oracle.h")
    if (not force and os.path.exists(out)
            and os.path.getmtime(out) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return out
    tmp = out + f".tmp{os.getpid()}"
    cmd = ["gcc", "-std=c11", *flags, "-fPIC", "-shared", "-o", tmp, src, "-lm"]
    subprocess.run(cmd, check=True)
    os.replace(tmp, out)
    return out


_i64 = ctypes.c_int64
_dp = ctypes.POINTER(ctypes.c_double)
_fp = ctypes.POINTER(ctypes.c_float)
PROGRESS_FN = ctypes.CFUNCTYPE(ctypes.c_int, _i64, ctypes.c_void_p)


def _d(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def _f(a):
    return a.ctypes.data_as(_fp) if a is not None else None


class Oracle:
    def __init__(self, flavour: str = "native"):
        self.path = build(flavour)
        self.lib = L = ctypes.CDLL(self.path)
        L.fo_window_hann.argtypes = [_i64, _dp]
        L.fo_stft_num_frames.argtypes = [_i64, _i64, _i64]
        L.fo_stft_num_frames.restype = _i64
        L.fo_stft.argtypes = [_dp, _i64, _i64, _i64, _i64, _dp, _dp]
        L.fo_stft.restype = _i64
        L.fo_stft_f32.argtypes = [_fp, _i64, _i64, _i64, _i64, _i64, _dp, _dp]
        L.fo_stft_f32.restype = _i64
        L.fo_rng_uniform01.argtypes = [ctypes.c_uint64, _i64, _dp]
        L.fo_nmf_process.argtypes = [_dp, _i64, _i64, _i64, _i64, ctypes.c_int, ctypes.c_int,
                                     _i64, _dp, _dp, _dp, _dp, _dp, ctypes.c_int, PROGRESS_FN,
                                     ctypes.c_void_p]
        L.fo_nmf_process.restype = ctypes.c_int
        L.fo_nmf_process_frame.argtypes = [_dp, _dp, _i64, _i64, _i64, _i64, _dp, _dp]
        L.fo_nmf_process_frame.restype = None
        L.fo_bufnmf_writeback.argtypes = [_dp, _dp, _i64, _i64, _i64, _fp, _fp]
        L.fo_bufnmf_channel.argtypes = [_fp, _i64, _i64, _i64, _i64, _i64, _i64, _i64,
                                        ctypes.c_int, _fp, _fp, _dp]
        L.fo_bufnmf_channel.restype = _i64
        L.fo_resynth_component.argtypes = [_dp, _dp, _dp, _dp, _i64, _i64, _i64, _i64, _i64,
                                           _i64, _i64, _i64, _dp]
        _dbl = ctypes.c_double
        L.fo_mel_filters.argtypes = [_dbl, _dbl, _i64, _i64, _dbl, _dp]
        L.fo_dct_table.argtypes = [_i64, _i64, _dp]
        L.fo_bufmfcc_channel.argtypes = [_fp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _dbl, _dbl, _dbl, _fp]
        L.fo_bufmfcc_channel.restype = _i64
        L.fo_bufmelbands_channel.argtypes = [_fp, _i64, _i64, _i64, _i64, _i64, _dbl, _dbl, _dbl,
                                             ctypes.c_int, ctypes.c_int, _fp]
        L.fo_bufmelbands_channel.restype = _i64
        L.fo_bufstft_num_hops.argtypes = [_i64, _i64, _i64, ctypes.c_int]
        L.fo_bufstft_num_hops.restype = _i64
        L.fo_bufstft_forward.argtypes = [_fp, _i64, _i64, _i64, _i64, ctypes.c_int, _fp, _fp]
        L.fo_bufstft_forward.restype = _i64
        L.fo_bufstft_inverse.argtypes = [_fp, _fp, _i64, _i64, _i64, _i64, ctypes.c_int, _dp]
        L.fo_bufstft_inverse.restype = _i64

    # ---- wrappers returning numpy arrays ------------------------------------------------
    def hann(self, win):
        w = np.empty(win)
        self.lib.fo_window_hann(win, _d(w))
        return w

    def num_frames(self, n, win, hop):
        return int(self.lib.fo_stft_num_frames(n, win, hop))

    def stft(self, audio, win, fft, hop):
        audio = np.ascontiguousarray(audio, dtype=np.float64)
        n = audio.shape[0]
        T, F = self.num_frames(n, win, hop), fft // 2 + 1
        spec = np.empty((T, F, 2))
        mag = np.empty((T, F))
        Tr = self.lib.fo_stft(_d(audio), n, win, fft, hop, _d(spec), _d(mag))
        assert Tr == T
        return spec[..., 0] + 1j * spec[..., 1], mag

    def stft_f32(self, audio, win, fft, hop):
        audio = np.ascontiguousarray(audio, dtype=np.float32)
        n = audio.shape[0]
        T, F = self.num_frames(n, win, hop), fft // 2 + 1
        spec = np.empty((T, F, 2))
        mag = np.empty((T, F))
        self.lib.fo_stft_f32(_f(audio), n, 1, win, fft, hop, _d(spec), _d(mag))
        return spec[..., 0] + 1j * spec[..., 1], mag

    def fma_peak_gflops(self, seconds=0.2):
        """the host core's FMA rate as this build reaches it with nothing but register FMAs in the loop (best of three)"""
        import time
        fpr = _i64(0)
        self.lib.fo_fma_burst.restype = ctypes.c_double
        self.lib.fo_fma_burst.argtypes = [_i64, ctypes.POINTER(_i64)]
        reps, best = 1 << 20, 0.0
        for _ in range(4):
            t0 = time.perf_counter()
            self.lib.fo_fma_burst(reps, ctypes.byref(fpr))
            dt = time.perf_counter() - t0
            best = max(best, reps * fpr.value / dt / 1e9)
            reps = int(max(1 << 18, reps * (seconds / 3) / max(dt, 1e-6)))
        return best

    def rng_uniform01(self, seed, count):
        out = np.empty(count)
        self.lib.fo_rng_uniform01(seed, count, _d(out))
        return out

    def nmf_process(self, X, K, iters, updateW=True, updateH=True, seed=42, W0=None, H0=None,
                    faithful=False, progress=None):
        X = np.ascontiguousarray(X, dtype=np.float64)
        T, F = X.shape
        W0c = None if W0 is None else np.ascontiguousarray(W0, dtype=np.float64)
        H0c = None if H0 is None else np.ascontiguousarray(H0, dtype=np.float64)
        W1, H1, V1 = np.empty((K, F)), np.empty((T, K)), np.empty((T, F))
        cb = PROGRESS_FN(lambda it, _u: 1 if progress(int(it)) else 0) if progress else PROGRESS_FN()
        rc = self.lib.fo_nmf_process(_d(X), T, F, K, iters, int(updateW), int(updateH), seed,
                                     _d(W0c), _d(H0c), _d(W1), _d(H1), _d(V1), int(faithful),
                                     cb, None)
        return W1, H1, V1, rc

    def nmf_process_frames(self, X, W0, iters, seed):
        """alg/NMF.hpp:45-89 on every row of X [T,F] with the dictionary W0 [K,F]: H [T,K], V [T,F]."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        W0 = np.ascontiguousarray(W0, dtype=np.float64)
        T, F = X.shape
        K = W0.shape[0]
        H, V = np.empty((T, K)), np.empty((T, F))
        for t in range(T):
            self.lib.fo_nmf_process_frame(_d(X[t]), _d(W0), K, F, iters, seed, _d(H[t]), _d(V[t]))
        return H, V

    def bufnmf_writeback(self, W1, H1):
        K, F = W1.shape
        T = H1.shape[0]
        bases = np.empty((K, F), dtype=np.float32)
        acts = np.empty((K, T), dtype=np.float32)
        self.lib.fo_bufnmf_writeback(_d(np.ascontiguousarray(W1)), _d(np.ascontiguousarray(H1)),
                                     T, F, K, _f(bases), _f(acts))
        return bases, acts

    def bufnmf_channel(self, audio, win, fft, hop, K, iters, seed, faithful=False,
                       want_mag=False):
        audio = np.ascontiguousarray(audio, dtype=np.float32)
        n = audio.shape[0]
        T, F = self.num_frames(n, win, hop), fft // 2 + 1
        bases = np.empty((K, F), dtype=np.float32)
        acts = np.empty((K, T), dtype=np.float32)
        mag = np.empty((T, F)) if want_mag else None
        self.lib.fo_bufnmf_channel(_f(audio), n, win, fft, hop, K, iters, seed, int(faithful),
                                   _f(bases), _f(acts), _d(mag))
        return (bases, acts, mag) if want_mag else (bases, acts)

    def resynth_component(self, spec, W1, H1, V1, k, win, fft, hop, n):
        T, F = spec.shape
        K = W1.shape[0]
        s = np.empty((T, F, 2))
        s[..., 0], s[..., 1] = spec.real, spec.imag
        out = np.empty(n)
        self.lib.fo_resynth_component(_d(s), _d(np.ascontiguousarray(W1)),
                                      _d(np.ascontiguousarray(H1)),
                                      _d(np.ascontiguousarray(V1)), T, F, K, k, win, fft, hop, n,
                                      _d(out))
        return out


    def _feature_T(self, n, win, hop):
        return 1 + (n + 2 * (win // 2)) // hop - win // hop

    def mel_filters(self, lo, hi, n_bands, n_bins, sr):
        f = np.empty((n_bands, n_bins))
        self.lib.fo_mel_filters(lo, hi, n_bands, n_bins, sr, _d(f))
        return f

    def dct_table(self, n_in, n_out):
        t = np.empty((n_out, n_in))
        self.lib.fo_dct_table(n_in, n_out, _d(t))
        return t

    def bufmfcc_channel(self, audio, win, fft, hop, n_bands=40, n_coefs=13, start_coeff=0, lo=20.0,
                        hi=20000.0, sr=44100.0):
        audio = np.ascontiguousarray(audio, dtype=np.float32)
        n = audio.shape[0]
        T = self._feature_T(n, win, hop)
        out = np.empty((n_coefs, T), dtype=np.float32)
        Tr = self.lib.fo_bufmfcc_channel(_f(audio), n, win, fft, hop, n_bands, n_coefs, start_coeff, lo, hi, sr, _f(out))
        assert Tr == T
        return out

    def bufmelbands_channel(self, audio, win, fft, hop, n_bands=40, lo=20.0, hi=20000.0, sr=44100.0,
                            normalize=True, scale_db=False):
        audio = np.ascontiguousarray(audio, dtype=np.float32)
        n = audio.shape[0]
        T = self._feature_T(n, win, hop)
        out = np.empty((n_bands, T), dtype=np.float32)
        Tr = self.lib.fo_bufmelbands_channel(_f(audio), n, win, fft, hop, n_bands, lo, hi, sr, int(normalize),
                                             int(scale_db), _f(out))
        assert Tr == T
        return out


    def bufstft_forward(self, audio, win, fft, hop, padding_mode=1):
        audio = np.ascontiguousarray(audio, dtype=np.float32)
        n = audio.shape[0]
        T, F = int(self.lib.fo_bufstft_num_hops(n, win, hop, padding_mode)), fft // 2 + 1
        mag = np.empty((F, T), dtype=np.float32)
        ph = np.empty((F, T), dtype=np.float32)
        self.lib.fo_bufstft_forward(_f(audio), n, win, fft, hop, padding_mode, _f(mag), _f(ph))
        return mag, ph

    def bufstft_inverse(self, mag, phase, win, fft, hop, padding_mode=1):
        mag = np.ascontiguousarray(mag, dtype=np.float32)
        phase = np.ascontiguousarray(phase, dtype=np.float32)
        F, T = mag.shape
        pad = [0, win >> 1, win - hop][padding_mode]
        out = np.empty((T - 1) * hop + win - pad)
        n = self.lib.fo_bufstft_inverse(_f(mag), _f(phase), T, win, fft, hop, padding_mode, _d(out))
        assert n == out.shape[0]
        return out


_cache = {}


def get(flavour: str = "native") -> Oracle:
    if flavour not in _cache:
        _cache[flavour] = Oracle(flavour)
    return _cache[flavour]
