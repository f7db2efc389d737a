"""ctypes binding of libflucoma_hip.so (include/flucoma_hip.h) for the Python-side tests and
bench.py.  The product is the shared library and the C++ client headers; this module is a thin
caller that mirrors the C ABI one to one and FAILS LOUDLY when the library is missing -- there
is no numpy / torch fallback anywhere.
"""
from __future__ import annotations

import ctypes
import weakref
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FLUHIP_LIB: another build of the library (A/B timing of two source states on one GPU box); FLUHIP_AB=1: the build with the
# experiment switches of DESIGN 6b compiled in (build.py build_ab) -- the production library does not read them
LIB_AB_PATH = os.path.join(_HERE, "lib_ab", "libflucoma_hip_ab.so")
LIB_QC_PATH = os.path.join(_HERE, "lib_ab", "libflucoma_hip_qc.so")
LIB_PATH = os.environ.get("FLUHIP_LIB") or (LIB_AB_PATH if os.environ.get("FLUHIP_AB") == "1" else
                                            os.path.join(_HERE, "lib", "libflucoma_hip.so"))

OK, WARNING, ERROR, CANCELLED = 0, 1, 2, 3

_i64 = ctypes.c_int64
_dp = ctypes.POINTER(ctypes.c_double)
_fp = ctypes.POINTER(ctypes.c_float)
_ip = ctypes.POINTER(ctypes.c_int64)
_vp = ctypes.c_void_p
PROGRESS_FN = ctypes.CFUNCTYPE(ctypes.c_int, _i64, ctypes.c_void_p)


class MatrixView(ctypes.Structure):
    """fluhip_matrix_view: element (r, c) at data[r * row_stride + c * col_stride] (strides in doubles)"""
    _fields_ = [("data", ctypes.POINTER(ctypes.c_double)), ("rows", _i64), ("cols", _i64), ("row_stride", _i64),
                ("col_stride", _i64)]

    @classmethod
    def of(cls, a):
        """view of a 2-D float64 numpy array with whatever strides it has (a.T, a[::2, 1:], ...)"""
        assert a.dtype == np.float64 and a.ndim == 2 and all(st % 8 == 0 for st in a.strides)
        return cls(ctypes.cast(a.ctypes.data, ctypes.POINTER(ctypes.c_double)), a.shape[0], a.shape[1],
                   a.strides[0] // 8, a.strides[1] // 8)

class BufNMFJob(ctypes.Structure):
    """fluhip_bufnmf_job"""
    _fields_ = [("count", _i64), ("n", _i64), ("win", _i64), ("fft", _i64), ("hop", _i64), ("K", _i64), ("iters", _i64),
                ("update_w", ctypes.c_int), ("update_h", ctypes.c_int), ("seed", _i64), ("seeds", _ip),
                ("audio", _fp), ("bases_seed", _fp), ("acts_seed", _fp), ("bases", _fp), ("acts", _fp), ("resynth", _fp)]


EXPORTS = [
    "fluhip_abi_version", "fluhip_device_count", "fluhip_ctx_create", "fluhip_ctx_destroy",
    "fluhip_last_error", "fluhip_ctx_device_info", "fluhip_ctx_stream", "fluhip_ctx_synchronize", "fluhip_ctx_trim", "fluhip_ctx_set_progress_lag",
    "fluhip_fft_params", "fluhip_stft_num_frames", "fluhip_stft_f64", "fluhip_stft_f32",
    "fluhip_nmf_process_f64", "fluhip_nmf_process_views_f64", "fluhip_nmf_process_frames_f64", "fluhip_nndsvd_f64", "fluhip_bufnmfseed_f32",
    "fluhip_bufnmf_channel_f32", "fluhip_bufmelbands_f32", "fluhip_bufmfcc_f32", "fluhip_bufmelbands_padded_f32",
    "fluhip_bufmfcc_padded_f32",
    "fluhip_bufstft_forward_f32", "fluhip_bufstft_inverse_f32",
    "fluhip_corpus_create", "fluhip_corpus_create_ragged", "fluhip_corpus_frames_of", "fluhip_corpus_set_audio_ragged_host",
    "fluhip_corpus_writeback_ragged_host", "fluhip_corpus_resynth_ragged_host",
    "fluhip_corpus_destroy", "fluhip_corpus_frames", "fluhip_corpus_bins",
    "fluhip_corpus_device_bytes", "fluhip_corpus_set_audio_host", "fluhip_corpus_set_audio_dev",
    "fluhip_corpus_stft", "fluhip_corpus_stft_mag_only", "fluhip_corpus_nmf", "fluhip_corpus_set_factors", "fluhip_corpus_writeback_dev",
    "fluhip_corpus_writeback_host", "fluhip_corpus_keep_spectrum", "fluhip_corpus_resynth_dev",
    "fluhip_corpus_resynth_host", "fluhip_corpus_resynth_interleaved_host", "fluhip_corpus_read_f64", "fluhip_corpus_plan", "fluhip_prof_enable",
    "fluhip_prof_reset", "fluhip_prof_read", "fluhip_corpus_debug_words", "fluhip_corpus_update_clocks", "fluhip_corpus_last_loop_ms", "fluhip_last_error_is_out_of_memory", "fluhip_clear_error", "fluhip_debug_plan_lists", "fluhip_debug_plan_tail", "fluhip_debug_plan_kind", "fluhip_debug_plan_h_update", "fluhip_debug_plan_shape", "fluhip_debug_wnorm_form",
    "fluhip_pool_create", "fluhip_pool_destroy", "fluhip_pool_size", "fluhip_pool_device", "fluhip_pool_last_error",
    "fluhip_pool_bufnmf_f32", "fluhip_pool_bufnmf_job_f32", "fluhip_pool_bufnmf_ragged_f32", "fluhip_pool_bufmfcc_f32",
    "fluhip_pool_bufmelbands_f32", "fluhip_shard_range", "fluhip_balanced_assignment", "fluhip_nmfmatch_f32", "fluhip_nmffilter_f32",
]


class FluhipError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"fluhip status {status}: {message}")
        self.status = status
        self.message = message


def load_library(path: str = LIB_PATH) -> ctypes.CDLL:
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} is missing: build it with `python flucoma-core_amd/build.py` "
            "(there is no fallback implementation)")
    L = ctypes.CDLL(path)
    L.fluhip_abi_version.restype = ctypes.c_int
    L.fluhip_device_count.restype = ctypes.c_int
    L.fluhip_ctx_create.argtypes = [ctypes.c_int, ctypes.POINTER(_vp)]
    L.fluhip_ctx_destroy.argtypes = [_vp]
    L.fluhip_ctx_destroy.restype = None
    L.fluhip_last_error.argtypes = [_vp]
    L.fluhip_last_error.restype = ctypes.c_char_p
    L.fluhip_ctx_device_info.argtypes = [_vp, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p,
                                         ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    L.fluhip_ctx_stream.argtypes = [_vp]
    L.fluhip_ctx_stream.restype = _vp
    L.fluhip_ctx_synchronize.argtypes = [_vp]
    L.fluhip_ctx_trim.argtypes = [_vp]
    L.fluhip_ctx_set_progress_lag.argtypes = [_vp, ctypes.c_int]
    L.fluhip_fft_params.argtypes = [_i64, _i64, _i64, _ip, _ip, _ip, _ip]
    L.fluhip_stft_num_frames.argtypes = [_i64, _i64, _i64]
    L.fluhip_stft_num_frames.restype = _i64
    L.fluhip_stft_f64.argtypes = [_vp, _dp, _i64, _i64, _i64, _i64, _i64, ctypes.c_int, _dp, _dp, _ip]
    L.fluhip_stft_f32.argtypes = [_vp, _fp, _i64, _i64, _i64, _i64, _i64, ctypes.c_int, _dp, _dp, _ip]
    L.fluhip_nmf_process_f64.argtypes = [_vp, _dp, _i64, _i64, _i64, _i64, _i64, ctypes.c_int,
                                         ctypes.c_int, _i64, _dp, _dp, _dp, _dp, _dp, PROGRESS_FN, _vp]
    _mv = ctypes.POINTER(MatrixView)
    L.fluhip_nmf_process_views_f64.argtypes = [_vp, _mv, _i64, _i64, ctypes.c_int, ctypes.c_int, _i64, _mv, _mv, _mv, _mv, _mv,
                                               PROGRESS_FN, _vp]
    L.fluhip_nmf_process_frames_f64.argtypes = [_vp, _dp, _i64, _i64, _i64, _dp, _i64, _i64, _i64, _dp, _dp]
    _dbl = ctypes.c_double
    L.fluhip_nndsvd_f64.argtypes = [_vp, _dp, _i64, _i64, _i64, _i64, _i64, _i64, _dbl, ctypes.c_int, _i64, _dp, _dp, _ip]
    L.fluhip_bufnmfseed_f32.argtypes = [_vp, _fp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _dbl, ctypes.c_int, _i64,
                                        _fp, _fp, _ip]
    L.fluhip_bufnmf_channel_f32.argtypes = [_vp, _fp, _i64, _i64, _i64, _i64, _i64, _i64, _i64,
                                            ctypes.c_int, ctypes.c_int, _i64, _fp, _fp, _fp, _fp,
                                            _fp, PROGRESS_FN, _vp]
    L.fluhip_bufmelbands_f32.argtypes = [_vp, _fp, _i64, _i64, _i64, _i64, _i64, _i64, _dbl, _dbl, _dbl,
                                         ctypes.c_int, ctypes.c_int, _fp, _ip]
    L.fluhip_bufmfcc_f32.argtypes = [_vp, _fp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _dbl, _dbl, _dbl,
                                     _fp, _ip]
    L.fluhip_bufmelbands_padded_f32.argtypes = [_vp, _fp, _i64, _i64, _i64, _i64, _i64, _i64, _dbl, _dbl, _dbl,
                                                ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, _ip]
    L.fluhip_bufmfcc_padded_f32.argtypes = [_vp, _fp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _dbl, _dbl, _dbl,
                                            ctypes.c_int, _fp, _ip]
    L.fluhip_bufstft_forward_f32.argtypes = [_vp, _fp, _i64, _i64, _i64, _i64, _i64, ctypes.c_int, _fp, _fp, _ip]
    L.fluhip_bufstft_inverse_f32.argtypes = [_vp, _fp, _fp, _i64, _i64, _i64, _i64, ctypes.c_int, _fp, _ip]
    L.fluhip_corpus_create.argtypes = [_vp, _i64, _i64, _i64, _i64, _i64, _i64, ctypes.POINTER(_vp)]
    L.fluhip_corpus_create_ragged.argtypes = [_vp, _i64, _ip, _i64, _i64, _i64, _i64, ctypes.POINTER(_vp)]
    L.fluhip_corpus_frames_of.argtypes = [_vp, _i64]
    L.fluhip_corpus_frames_of.restype = _i64
    L.fluhip_corpus_set_audio_ragged_host.argtypes = [_vp, ctypes.POINTER(_fp)]
    L.fluhip_corpus_writeback_ragged_host.argtypes = [_vp, ctypes.POINTER(_fp), ctypes.POINTER(_fp)]
    L.fluhip_corpus_resynth_ragged_host.argtypes = [_vp, ctypes.POINTER(_fp)]
    L.fluhip_corpus_destroy.argtypes = [_vp]
    L.fluhip_corpus_destroy.restype = None
    for f in ("fluhip_corpus_frames", "fluhip_corpus_bins", "fluhip_corpus_device_bytes"):
        getattr(L, f).argtypes = [_vp]
        getattr(L, f).restype = _i64
    L.fluhip_corpus_set_audio_host.argtypes = [_vp, _fp]
    L.fluhip_corpus_set_audio_dev.argtypes = [_vp, _vp]
    L.fluhip_corpus_stft.argtypes = [_vp]
    L.fluhip_corpus_stft_mag_only.argtypes = [_vp]
    L.fluhip_corpus_nmf.argtypes = [_vp, _i64, ctypes.c_int, ctypes.c_int, _i64, _ip, PROGRESS_FN, _vp]
    L.fluhip_corpus_set_factors.argtypes = [_vp, _fp, _fp]
    L.fluhip_corpus_writeback_dev.argtypes = [_vp, _vp, _vp]
    L.fluhip_corpus_writeback_host.argtypes = [_vp, _fp, _fp]
    L.fluhip_corpus_read_f64.argtypes = [_vp, _dp, _dp, _dp]
    L.fluhip_corpus_plan.argtypes = [_vp, _ip]
    L.fluhip_corpus_keep_spectrum.argtypes = [_vp, ctypes.c_int]
    L.fluhip_corpus_resynth_dev.argtypes = [_vp, _vp]
    L.fluhip_corpus_resynth_host.argtypes = [_vp, _fp]
    L.fluhip_corpus_resynth_interleaved_host.argtypes = [_vp, _fp, _i64]
    L.fluhip_prof_enable.argtypes = [_vp, ctypes.c_int]
    L.fluhip_prof_reset.argtypes = [_vp]
    L.fluhip_prof_read.argtypes = [_vp, ctypes.c_int, _ip, _dp]
    L.fluhip_corpus_debug_words.argtypes = [_vp, _ip]
    L.fluhip_corpus_update_clocks.argtypes = [_vp, _ip, ctypes.c_int]
    L.fluhip_last_error_is_out_of_memory.argtypes = [_vp]
    L.fluhip_last_error_is_out_of_memory.restype = ctypes.c_int
    L.fluhip_clear_error.argtypes = [_vp]
    L.fluhip_clear_error.restype = None
    L.fluhip_debug_plan_shape.argtypes = [_i64, _i64, _i64, _i64, _ip]
    L.fluhip_debug_wnorm_form.argtypes = [ctypes.c_int] * 7
    L.fluhip_corpus_last_loop_ms.argtypes = [_vp, ctypes.POINTER(ctypes.c_double)]
    L.fluhip_debug_plan_lists.argtypes = [_i64, _ip, _i64, _i64, ctypes.c_int, ctypes.POINTER(ctypes.c_int32), _i64,
                                          ctypes.POINTER(ctypes.c_int32)]
    L.fluhip_debug_plan_lists.restype = _i64
    L.fluhip_debug_plan_tail.argtypes = [_i64, _i64, _i64, _i64, _ip]
    L.fluhip_debug_plan_kind.argtypes = [_i64, _i64, _i64, _i64]
    L.fluhip_debug_plan_h_update.argtypes = [_i64, _i64, _i64, _i64]
    L.fluhip_pool_create.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.POINTER(_vp)]
    L.fluhip_pool_destroy.argtypes = [_vp]
    L.fluhip_pool_destroy.restype = None
    L.fluhip_pool_size.argtypes = [_vp]
    L.fluhip_pool_device.argtypes = [_vp, ctypes.c_int]
    L.fluhip_pool_last_error.argtypes = [_vp]
    L.fluhip_pool_last_error.restype = ctypes.c_char_p
    L.fluhip_pool_bufmfcc_f32.argtypes = [_vp, _fp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _dbl, _dbl, _dbl,
                                          ctypes.c_int, _fp, _ip]
    L.fluhip_pool_bufmelbands_f32.argtypes = [_vp, _fp, _i64, _i64, _i64, _i64, _i64, _i64, _dbl, _dbl, _dbl,
                                              ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, _ip]
    L.fluhip_pool_bufnmf_f32.argtypes = [_vp, _fp, _i64, _i64, _i64, _i64, _i64, _i64, _i64, ctypes.c_int, ctypes.c_int,
                                         _i64, _ip, _fp, _fp, PROGRESS_FN, _vp]
    L.fluhip_pool_bufnmf_job_f32.argtypes = [_vp, ctypes.POINTER(BufNMFJob), PROGRESS_FN, _vp]
    L.fluhip_pool_bufnmf_ragged_f32.argtypes = [_vp, ctypes.POINTER(_fp), _ip, _i64, _i64, _i64, _i64, _i64, _i64, ctypes.c_int,
                                                ctypes.c_int, _i64, _ip, ctypes.POINTER(_fp), ctypes.POINTER(_fp), PROGRESS_FN, _vp]
    L.fluhip_shard_range.argtypes = [_i64, ctypes.c_int, ctypes.c_int, _ip, _ip]
    L.fluhip_shard_range.restype = None
    L.fluhip_balanced_assignment.argtypes = [_dp, _i64, ctypes.c_int, ctypes.POINTER(ctypes.c_int32)]
    return L


def _d(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def _f(a):
    return a.ctypes.data_as(_fp) if a is not None else None


def _cb(progress):
    if progress is None:
        return PROGRESS_FN()
    return PROGRESS_FN(lambda it, _u: 1 if progress(int(it)) else 0)


class Context:
    """fluhip_ctx: one device, one stream."""

    def __init__(self, device: int = 0, lib: ctypes.CDLL | None = None):
        self.lib = lib or load_library()
        h = _vp()
        rc = self.lib.fluhip_ctx_create(device, ctypes.byref(h))
        if rc != OK:
            raise FluhipError(rc, f"cannot create context on device {device} "
                                  f"({self.lib.fluhip_device_count()} HIP devices visible)")
        self.h = h

    def close(self):
        if self.h:
            # a corpus must not outlive its context (its device buffers are freed on the context's stream)
            for ref in list(getattr(self, "_corpora", [])):
                c = ref()
                if c is not None:
                    c.close()
            self.lib.fluhip_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, allow=(OK,)):
        if rc not in allow:
            raise FluhipError(rc, self.lib.fluhip_last_error(self.h).decode())
        return rc

    def device_info(self):
        name = ctypes.create_string_buffer(256)
        arch = ctypes.create_string_buffer(256)
        cus = ctypes.c_int(0)
        self._check(self.lib.fluhip_ctx_device_info(self.h, name, 256, arch, 256, ctypes.byref(cus)))
        return name.value.decode(), arch.value.decode(), cus.value

    def synchronize(self):
        self._check(self.lib.fluhip_ctx_synchronize(self.h))

    def set_progress_lag(self, lag):
        self._check(self.lib.fluhip_ctx_set_progress_lag(self.h, int(lag)))

    # ---- algorithm::STFT ----------------------------------------------------------------
    def stft(self, audio, win, fft, hop, window_type=0, want_spec=True, want_mag=True, stride=1):
        audio = np.ascontiguousarray(audio)
        n = (audio.shape[0] + stride - 1) // stride
        T = (n + hop) // hop
        F = fft // 2 + 1
        spec = np.empty((T, F, 2)) if want_spec else None
        mag = np.empty((T, F)) if want_mag else None
        Tout = _i64(0)
        if audio.dtype == np.float32:
            rc = self.lib.fluhip_stft_f32(self.h, _f(audio), n, stride, win, fft, hop, window_type,
                                          _d(spec), _d(mag), ctypes.byref(Tout))
        else:
            audio = audio.astype(np.float64, copy=False)
            rc = self.lib.fluhip_stft_f64(self.h, _d(audio), n, stride, win, fft, hop, window_type,
                                          _d(spec), _d(mag), ctypes.byref(Tout))
        self._check(rc)
        assert Tout.value == T
        cs = spec[..., 0] + 1j * spec[..., 1] if want_spec else None
        return cs, mag

    # ---- algorithm::NMF -----------------------------------------------------------------
    def nmf_process(self, X, K, iters, updateW=True, updateH=True, seed=42, W0=None, H0=None,
                    progress=None, want_v=True):
        X = np.asarray(X, dtype=np.float64)
        assert X.ndim == 2 and X.strides[1] == 8
        T, F = X.shape
        ldx = X.strides[0] // 8
        W0c = None if W0 is None else np.ascontiguousarray(W0, dtype=np.float64)
        H0c = None if H0 is None else np.ascontiguousarray(H0, dtype=np.float64)
        W1, H1 = np.empty((K, F)), np.empty((T, K))
        V1 = np.empty((T, F)) if want_v else None
        cb = _cb(progress)
        rc = self.lib.fluhip_nmf_process_f64(self.h, X.ctypes.data_as(_dp), T, F, ldx, K, iters,
                                             int(updateW), int(updateH), seed, _d(W0c), _d(H0c),
                                             _d(W1), _d(H1), _d(V1), cb, None)
        self._check(rc, allow=(OK, CANCELLED))
        return W1, H1, V1, rc

    def nmf_process_views(self, X, K, iters, updateW=True, updateH=True, seed=42, W0=None, H0=None, W1=None, H1=None,
                          V1=None, progress=None):
        """NMF::process on numpy arrays used IN PLACE with their own strides (transposed views, sub-blocks)"""
        ref = lambda a: ctypes.byref(MatrixView.of(a)) if a is not None else None  # noqa: E731
        cb = _cb(progress)
        rc = self.lib.fluhip_nmf_process_views_f64(self.h, ref(X), K, iters, int(updateW), int(updateH), seed, ref(W0),
                                                   ref(H0), ref(W1), ref(H1), ref(V1), cb, None)
        return self._check(rc, allow=(OK, CANCELLED))

    def nmf_process_frames(self, X, W0, iters, seed=42, want_v=True):
        """NMF::processFrame (alg/NMF.hpp:45-89) on every row of X [T,F] with the dictionary W0 [K,F]."""
        X = np.asarray(X, dtype=np.float64)
        assert X.ndim == 2 and X.strides[1] == 8
        T, F = X.shape
        W0c = np.ascontiguousarray(W0, dtype=np.float64)
        K = W0c.shape[0]
        assert W0c.shape == (K, F)
        H = np.empty((T, K))
        V = np.empty((T, F)) if want_v else None
        self._check(self.lib.fluhip_nmf_process_frames_f64(self.h, X.ctypes.data_as(_dp), T, F, X.strides[0] // 8,
                                                           _d(W0c), K, iters, seed, _d(H), _d(V)))
        return H, V

    def nndsvd(self, X, w_rows, min_rank=0, max_rank=200, amount=0.8, method=0, seed=-1):
        """NNDSVD::process (alg/NNDSVD.hpp:30-132): W [w_rows,F], H [T,w_rows], rank."""
        X = np.asarray(X, dtype=np.float64)
        assert X.ndim == 2 and X.strides[1] == 8
        T, F = X.shape
        W, H = np.empty((w_rows, F)), np.empty((T, w_rows))
        k = ctypes.c_int64(0)
        self._check(self.lib.fluhip_nndsvd_f64(self.h, X.ctypes.data_as(_dp), T, F, X.strides[0] // 8, w_rows, min_rank,
                                               max_rank, amount, method, seed, _d(W), _d(H), ctypes.byref(k)))
        return W, H, int(k.value)

    def bufnmfseed(self, audio, win, fft, hop, min_rank=1, max_rank=200, coverage=0.5, method=0, seed=-1, stride=1):
        """BufNMFSeed (nrt/NMFSeedClient.hpp): bases [max_rank,F] f32, activations [max_rank,T] f32, rank."""
        audio = np.ascontiguousarray(audio, dtype=np.float32)
        n = (audio.shape[0] + stride - 1) // stride
        T, F = (n + hop) // hop, fft // 2 + 1
        bases = np.empty((max_rank, F), dtype=np.float32)
        acts = np.empty((max_rank, T), dtype=np.float32)
        k = ctypes.c_int64(0)
        self._check(self.lib.fluhip_bufnmfseed_f32(self.h, _f(audio), n, stride, win, fft, hop, min_rank, max_rank,
                                                   coverage, method, seed, _f(bases), _f(acts), ctypes.byref(k)))
        return bases, acts, int(k.value)

    # ---- one BufNMF channel -------------------------------------------------------------
    def bufnmf_channel(self, audio, win, fft, hop, K, iters, seed, updateW=True, updateH=True,
                       bases_seed=None, acts_seed=None, progress=None, stride=1, resynth=False):
        audio = np.ascontiguousarray(audio, dtype=np.float32)
        n = (audio.shape[0] + stride - 1) // stride
        T, F = (n + hop) // hop, fft // 2 + 1
        bases = np.empty((K, F), dtype=np.float32)
        acts = np.empty((K, T), dtype=np.float32)
        bs = None if bases_seed is None else np.ascontiguousarray(bases_seed, dtype=np.float32)
        as_ = None if acts_seed is None else np.ascontiguousarray(acts_seed, dtype=np.float32)
        cb = _cb(progress)
        res = np.empty((K, n), dtype=np.float32) if resynth else None
        rc = self.lib.fluhip_bufnmf_channel_f32(self.h, _f(audio), n, stride, win, fft, hop, K, iters,
                                                int(updateW), int(updateH), seed, _f(bs), _f(as_),
                                                _f(bases), _f(acts), _f(res), cb, None)
        self._check(rc, allow=(OK, CANCELLED))
        if resynth:
            return bases, acts, res, rc
        return bases, acts, rc

    # ---- feature pipeline -----------------------------------------------------------------
    @staticmethod
    def feature_frames(n, win, hop, padding_mode=1):
        pad = (0, win // 2, win - hop)[padding_mode]
        padded = n + win + 2 * pad
        if padding_mode == 2:
            padded = -(-padded // hop) * hop
        return 1 + (padded - win) // hop - win // hop

    def bufmfcc(self, audio, win, fft, hop, n_bands=40, n_coefs=13, start_coeff=0, lo=20.0, hi=20000.0,
                sr=44100.0, padding_mode=1):
        audio = np.ascontiguousarray(np.atleast_2d(audio), dtype=np.float32)
        count, n = audio.shape
        T = self.feature_frames(n, win, hop, padding_mode)
        out = np.empty((count, n_coefs, T), dtype=np.float32)
        Tr = _i64(0)
        if padding_mode == 1:
            rc = self.lib.fluhip_bufmfcc_f32(self.h, _f(audio), count, n, win, fft, hop, n_bands, n_coefs,
                                             start_coeff, lo, hi, sr, _f(out), ctypes.byref(Tr))
        else:
            rc = self.lib.fluhip_bufmfcc_padded_f32(self.h, _f(audio), count, n, win, fft, hop, n_bands, n_coefs,
                                                    start_coeff, lo, hi, sr, padding_mode, _f(out), ctypes.byref(Tr))
        self._check(rc)
        assert Tr.value == T
        return out

    def bufmelbands(self, audio, win, fft, hop, n_bands=40, lo=20.0, hi=20000.0, sr=44100.0, normalize=True,
                    scale_db=False, padding_mode=1):
        audio = np.ascontiguousarray(np.atleast_2d(audio), dtype=np.float32)
        count, n = audio.shape
        T = self.feature_frames(n, win, hop, padding_mode)
        out = np.empty((count, n_bands, T), dtype=np.float32)
        Tr = _i64(0)
        if padding_mode == 1:
            rc = self.lib.fluhip_bufmelbands_f32(self.h, _f(audio), count, n, win, fft, hop, n_bands, lo, hi, sr,
                                                 int(normalize), int(scale_db), _f(out), ctypes.byref(Tr))
        else:
            rc = self.lib.fluhip_bufmelbands_padded_f32(self.h, _f(audio), count, n, win, fft, hop, n_bands, lo, hi, sr,
                                                        int(normalize), int(scale_db), padding_mode, _f(out),
                                                        ctypes.byref(Tr))
        self._check(rc)
        assert Tr.value == T
        return out

    # ---- BufSTFT ----------------------------------------------------------------------------
    def nmfmatch(self, audio, bases, win, fft, hop, seed=42, padding_mode=1):
        """NMFMatch over a buffer (fluhip_nmfmatch_f32): audio [channels, n] or [n] floats, bases [K, F] -> [channels, K, T]"""
        a = np.ascontiguousarray(np.atleast_2d(audio), dtype=np.float32)
        b = np.ascontiguousarray(bases, dtype=np.float32)
        count, n = a.shape
        K = b.shape[0]
        T = _i64(0)
        f = self.lib.fluhip_nmfmatch_f32
        f.argtypes = [_vp, _fp, _i64, _i64, _i64, _i64, _i64, _fp, _i64, _i64, ctypes.c_int, _fp, _ip]
        self._check(f(self.h, _f(a), count, n, win, fft, hop, _f(b), K, seed, padding_mode, None, ctypes.byref(T)))
        out = np.empty((count, K, T.value), dtype=np.float32)
        self._check(f(self.h, _f(a), count, n, win, fft, hop, _f(b), K, seed, padding_mode, _f(out), ctypes.byref(T)))
        return out

    def nmffilter(self, audio, bases, win, fft, hop, iters=10, seed=42):
        """NMFFilter over a buffer (fluhip_nmffilter_f32): -> [channels, K, n] floats"""
        a = np.ascontiguousarray(np.atleast_2d(audio), dtype=np.float32)
        b = np.ascontiguousarray(bases, dtype=np.float32)
        count, n = a.shape
        K = b.shape[0]
        out = np.empty((count, K, n), dtype=np.float32)
        f = self.lib.fluhip_nmffilter_f32
        f.argtypes = [_vp, _fp, _i64, _i64, _i64, _i64, _i64, _fp, _i64, _i64, _i64, _fp]
        self._check(f(self.h, _f(a), count, n, win, fft, hop, _f(b), K, iters, seed, _f(out)))
        return out

    def bufstft_forward(self, audio, win, fft, hop, padding_mode=1):
        audio = np.ascontiguousarray(audio, dtype=np.float32)
        n = audio.shape[0]
        pad = [0, win >> 1, win - hop][padding_mode]
        padded = n + 2 * pad
        if padding_mode == 2:
            padded = -(-padded // hop) * hop
        T, F = 1 + (padded - win) // hop, fft // 2 + 1
        mag = np.empty((F, T), dtype=np.float32)
        ph = np.empty((F, T), dtype=np.float32)
        Tr = _i64(0)
        self._check(self.lib.fluhip_bufstft_forward_f32(self.h, _f(audio), n, 1, win, fft, hop, padding_mode,
                                                        _f(mag), _f(ph), ctypes.byref(Tr)))
        assert Tr.value == T
        return mag, ph

    def bufstft_inverse(self, mag, phase, win, fft, hop, padding_mode=1):
        mag = np.ascontiguousarray(mag, dtype=np.float32)
        phase = np.ascontiguousarray(phase, dtype=np.float32)
        F, T = mag.shape
        nout = _i64(0)
        self._check(self.lib.fluhip_bufstft_inverse_f32(self.h, _f(mag), _f(phase), T, win, fft, hop, padding_mode,
                                                        None, ctypes.byref(nout)))
        out = np.empty(nout.value, dtype=np.float32)
        self._check(self.lib.fluhip_bufstft_inverse_f32(self.h, _f(mag), _f(phase), T, win, fft, hop, padding_mode,
                                                        _f(out), ctypes.byref(nout)))
        return out

    # ---- profiling ----------------------------------------------------------------------
    def prof_enable(self, on=True):
        self._check(self.lib.fluhip_prof_enable(self.h, int(on)))

    def prof_reset(self):
        self._check(self.lib.fluhip_prof_reset(self.h))

    def prof_read(self, cls):
        n, ms = _i64(0), ctypes.c_double(0)
        self._check(self.lib.fluhip_prof_read(self.h, cls, ctypes.byref(n), ctypes.byref(ms)))
        return n.value, ms.value


class Corpus:
    """fluhip_corpus: `count` equal-shape mono buffers resident in HBM."""

    def __init__(self, ctx: Context, count, n, win, fft, hop, K):
        self.ctx = ctx
        self.count, self.n, self.win, self.fft, self.hop, self.K = count, n, win, fft, hop, K
        h = _vp()
        ctx._check(ctx.lib.fluhip_corpus_create(ctx.h, count, n, win, fft, hop, K, ctypes.byref(h)))
        self.h = h
        if not hasattr(ctx, "_corpora"):
            ctx._corpora = []
        ctx._corpora.append(weakref.ref(self))
        self.T = int(ctx.lib.fluhip_corpus_frames(h))
        self.F = int(ctx.lib.fluhip_corpus_bins(h))

    def close(self):
        if self.h:
            if self.ctx.h:
                self.ctx.lib.fluhip_corpus_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_bytes(self):
        return int(self.ctx.lib.fluhip_corpus_device_bytes(self.h))

    def set_audio(self, audio):
        audio = np.ascontiguousarray(audio, dtype=np.float32)
        assert audio.shape == (self.count, self.n)
        self.ctx._check(self.ctx.lib.fluhip_corpus_set_audio_host(self.h, _f(audio)))

    def set_audio_dev(self, dev_ptr: int):
        self.ctx._check(self.ctx.lib.fluhip_corpus_set_audio_dev(self.h, _vp(dev_ptr)))

    def stft(self):
        self.ctx._check(self.ctx.lib.fluhip_corpus_stft(self.h))

    def stft_mag_only(self):
        """the frame-major magnitudes alone (what a spectrogram-only caller runs); nmf() then needs stft() again"""
        self.ctx._check(self.ctx.lib.fluhip_corpus_stft_mag_only(self.h))

    def nmf(self, iters, seed=42, updateW=True, updateH=True, seeds=None, progress=None):
        sarr = None if seeds is None else np.ascontiguousarray(seeds, dtype=np.int64)
        sp = sarr.ctypes.data_as(_ip) if sarr is not None else None
        cb = _cb(progress)
        rc = self.ctx.lib.fluhip_corpus_nmf(self.h, iters, int(updateW), int(updateH), seed, sp, cb, None)
        return self.ctx._check(rc, allow=(OK, CANCELLED))

    def set_factors(self, bases_seed=None, acts_seed=None):
        """Seed / Fixed factors for the following nmf() calls: [count, K, F] / [count, K, T] floats, or None for random draws"""
        bs = None if bases_seed is None else np.ascontiguousarray(bases_seed, dtype=np.float32)
        hs = None if acts_seed is None else np.ascontiguousarray(acts_seed, dtype=np.float32)
        assert bs is None or bs.shape == (self.count, self.K, self.F)
        assert hs is None or hs.shape == (self.count, self.K, self.T)
        self.ctx._check(self.ctx.lib.fluhip_corpus_set_factors(self.h, _f(bs), _f(hs)))

    def writeback_dev(self, bases_ptr: int | None, acts_ptr: int | None):
        self.ctx._check(self.ctx.lib.fluhip_corpus_writeback_dev(
            self.h, _vp(bases_ptr) if bases_ptr else None, _vp(acts_ptr) if acts_ptr else None))

    def writeback(self):
        bases = np.empty((self.count, self.K, self.F), dtype=np.float32)
        acts = np.empty((self.count, self.K, self.T), dtype=np.float32)
        self.ctx._check(self.ctx.lib.fluhip_corpus_writeback_host(self.h, _f(bases), _f(acts)))
        return bases, acts

    def keep_spectrum(self, on=True):
        self.ctx._check(self.ctx.lib.fluhip_corpus_keep_spectrum(self.h, int(on)))

    def resynth(self):
        """[count, K, n] f32: every component of every buffer resynthesised (NMFClient.hpp:302-334)"""
        out = np.empty((self.count, self.K, self.n), dtype=np.float32)
        self.ctx._check(self.ctx.lib.fluhip_corpus_resynth_host(self.h, _f(out)))
        return out

    def resynth_interleaved(self, frame_stride=None):
        """fluhip_corpus_resynth_interleaved_host: [n][frame_stride] floats, component k of buffer b in column b K + k"""
        chans = self.count * self.K
        fs = chans if frame_stride is None else frame_stride
        out = np.zeros((self.n, fs), dtype=np.float32)
        self.ctx._check(self.ctx.lib.fluhip_corpus_resynth_interleaved_host(self.h, _f(out), fs))
        return out

    def plan(self):
        out = (ctypes.c_int64 * 8)()
        self.ctx._check(self.ctx.lib.fluhip_corpus_plan(self.h, out))
        keys = ("kernel", "split_w", "split_h", "deferred_norm", "side_column", "strips_w", "padded_rank", "strip")
        d = dict(zip(keys, [int(v) for v in out]))
        d["compute_rank"] = d["padded_rank"] >> 16   # rank the factor updates compute (48 / 96 for the off-size ranks)
        d["padded_rank"] &= 0xFFFF
        d["tail_h"] = d["split_h"] >> 16  # pieces of the tail launch of a two-launch H update (0: one launch)
        d["split_h"] &= 0xFFFF
        return d

    def update_clocks(self, reset=False):
        """per factor update: launches, shader cycles and 100 MHz ticks of one wavefront per launch, summed since the last reset;
        derived: cycles per launch and the clock the part sustained (fluhip_corpus_update_clocks)"""
        out = (ctypes.c_int64 * 8)()
        self.ctx._check(self.ctx.lib.fluhip_corpus_update_clocks(self.h, out, int(reset)))
        res = {}
        for name, o in (("w", 0), ("h", 4)):
            n, cyc, ticks = int(out[o]), int(out[o + 1]), int(out[o + 2])
            res[name] = {"launches": n, "shader_cycles": cyc, "ticks_100mhz": ticks,
                         "cycles_per_launch": cyc / n if n else None,
                         "sustained_mhz": 100.0 * cyc / ticks if ticks else None}
        return res

    def last_loop_ms(self):
        """device time of the last nmf() call's iteration loop (HIP events behind the initialisation / the last launch)"""
        ms = ctypes.c_double(0)
        self.ctx._check(self.ctx.lib.fluhip_corpus_last_loop_ms(self.h, ctypes.byref(ms)))
        return ms.value

    def read_f64(self, mag=True, factors=True):
        m = np.empty((self.count, self.T, self.F)) if mag else None
        W1 = np.empty((self.count, self.K, self.F)) if factors else None
        H1 = np.empty((self.count, self.T, self.K)) if factors else None
        self.ctx._check(self.ctx.lib.fluhip_corpus_read_f64(self.h, _d(m), _d(W1), _d(H1)))
        return m, W1, H1


class RaggedCorpus(Corpus):
    """fluhip_corpus_create_ragged: mono buffers of DIFFERENT lengths as one device-resident batch.  T / n are those of the
    longest buffer (the shapes read_f64 returns; frames past a buffer's own are zero); Ts holds every buffer's own frames."""

    def __init__(self, ctx: Context, lens, win, fft, hop, K):
        self.ctx = ctx
        self.lens = [int(x) for x in lens]
        self.count, self.n, self.win, self.fft, self.hop, self.K = len(self.lens), max(self.lens), win, fft, hop, K
        arr = (ctypes.c_int64 * self.count)(*self.lens)
        h = _vp()
        ctx._check(ctx.lib.fluhip_corpus_create_ragged(ctx.h, self.count, arr, win, fft, hop, K, ctypes.byref(h)))
        self.h = h
        if not hasattr(ctx, "_corpora"):
            ctx._corpora = []
        ctx._corpora.append(weakref.ref(self))
        self.T = int(ctx.lib.fluhip_corpus_frames(h))
        self.F = int(ctx.lib.fluhip_corpus_bins(h))
        self.Ts = [int(ctx.lib.fluhip_corpus_frames_of(h, i)) for i in range(self.count)]

    def set_audio(self, audios):
        audios = [np.ascontiguousarray(a, dtype=np.float32) for a in audios]
        assert [a.shape[0] for a in audios] == self.lens
        ap = (_fp * self.count)(*[a.ctypes.data_as(_fp) for a in audios])
        self.ctx._check(self.ctx.lib.fluhip_corpus_set_audio_ragged_host(self.h, ap))

    def resynth(self):
        out = [np.empty((self.K, n), dtype=np.float32) for n in self.lens]
        op = (_fp * self.count)(*[o.ctypes.data_as(_fp) for o in out])
        self.ctx._check(self.ctx.lib.fluhip_corpus_resynth_ragged_host(self.h, op))
        return out

    def writeback(self):
        bases = [np.empty((self.K, self.F), dtype=np.float32) for _ in range(self.count)]
        acts = [np.empty((self.K, T), dtype=np.float32) for T in self.Ts]
        bp = (_fp * self.count)(*[b.ctypes.data_as(_fp) for b in bases])
        cp = (_fp * self.count)(*[c.ctypes.data_as(_fp) for c in acts])
        self.ctx._check(self.ctx.lib.fluhip_corpus_writeback_ragged_host(self.h, bp, cp))
        return bases, acts


class Pool:
    """fluhip_pool: several devices (or several contexts on one) behind one host process."""

    def __init__(self, devices=None, lib: ctypes.CDLL | None = None):
        self.lib = lib or load_library()
        h = _vp()
        if devices is None:
            rc = self.lib.fluhip_pool_create(None, 0, ctypes.byref(h))
        else:
            arr = (ctypes.c_int * len(devices))(*devices)
            rc = self.lib.fluhip_pool_create(arr, len(devices), ctypes.byref(h))
        if rc != OK:
            raise FluhipError(rc, "fluhip_pool_create failed (no usable device?)")
        self.h = h

    def close(self):
        if self.h:
            self.lib.fluhip_pool_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size(self):
        return int(self.lib.fluhip_pool_size(self.h))

    def devices(self):
        return [int(self.lib.fluhip_pool_device(self.h, i)) for i in range(self.size())]

    def bufnmf(self, audio, win, fft, hop, K, iters, seed=42, updateW=True, updateH=True, seeds=None, progress=None):
        audio = np.ascontiguousarray(audio, dtype=np.float32)
        count, n = audio.shape
        F, T = fft // 2 + 1, int(self.lib.fluhip_stft_num_frames(n, win, hop))
        bases = np.empty((count, K, F), dtype=np.float32)
        acts = np.empty((count, K, T), dtype=np.float32)
        sarr = None if seeds is None else np.ascontiguousarray(seeds, dtype=np.int64)
        cb = _cb(progress)
        rc = self.lib.fluhip_pool_bufnmf_f32(self.h, _f(audio), count, n, win, fft, hop, K, iters, int(updateW), int(updateH),
                                             seed, sarr.ctypes.data_as(_ip) if sarr is not None else None, _f(bases), _f(acts),
                                             cb, None)
        if rc not in (OK, CANCELLED):
            raise FluhipError(rc, self.lib.fluhip_pool_last_error(self.h).decode())
        return bases, acts, rc

    def bufmfcc(self, audio, win, fft, hop, n_bands=40, n_coefs=13, start_coeff=0, lo=20.0, hi=20000.0, sr=44100.0,
                padding_mode=1):
        """fluhip_pool_bufmfcc_f32: BASELINE config 5 over several devices, slices dealt in contiguous blocks"""
        audio = np.ascontiguousarray(np.atleast_2d(audio), dtype=np.float32)
        count, n = audio.shape
        T = Context.feature_frames(n, win, hop, padding_mode)
        out = np.empty((count, n_coefs, T), dtype=np.float32)
        Tr = _i64(0)
        rc = self.lib.fluhip_pool_bufmfcc_f32(self.h, _f(audio), count, n, win, fft, hop, n_bands, n_coefs, start_coeff, lo, hi, sr,
                                              padding_mode, _f(out), ctypes.byref(Tr))
        if rc != OK:
            raise FluhipError(rc, self.lib.fluhip_pool_last_error(self.h).decode())
        assert Tr.value == T
        return out

    def bufmelbands(self, audio, win, fft, hop, n_bands=40, lo=20.0, hi=20000.0, sr=44100.0, normalize=True, scale_db=False,
                    padding_mode=1):
        audio = np.ascontiguousarray(np.atleast_2d(audio), dtype=np.float32)
        count, n = audio.shape
        T = Context.feature_frames(n, win, hop, padding_mode)
        out = np.empty((count, n_bands, T), dtype=np.float32)
        Tr = _i64(0)
        rc = self.lib.fluhip_pool_bufmelbands_f32(self.h, _f(audio), count, n, win, fft, hop, n_bands, lo, hi, sr, int(normalize),
                                                  int(scale_db), padding_mode, _f(out), ctypes.byref(Tr))
        if rc != OK:
            raise FluhipError(rc, self.lib.fluhip_pool_last_error(self.h).decode())
        assert Tr.value == T
        return out

    def bufnmf_job(self, audio, win, fft, hop, K, iters, seed=42, updateW=True, updateH=True, seeds=None, bases_seed=None,
                   acts_seed=None, resynth=False, progress=None):
        """fluhip_pool_bufnmf_job_f32: the batched form with Seed / Fixed factors and the resynthesis output"""
        audio = np.ascontiguousarray(audio, dtype=np.float32)
        count, n = audio.shape
        F, T = fft // 2 + 1, int(self.lib.fluhip_stft_num_frames(n, win, hop))
        bases = np.empty((count, K, F), dtype=np.float32)
        acts = np.empty((count, K, T), dtype=np.float32)
        res = np.empty((count, K, n), dtype=np.float32) if resynth else None
        sarr = None if seeds is None else np.ascontiguousarray(seeds, dtype=np.int64)
        bs = None if bases_seed is None else np.ascontiguousarray(bases_seed, dtype=np.float32)
        hs = None if acts_seed is None else np.ascontiguousarray(acts_seed, dtype=np.float32)
        job = BufNMFJob(count, n, win, fft, hop, K, iters, int(updateW), int(updateH), seed,
                        sarr.ctypes.data_as(_ip) if sarr is not None else None, _f(audio), _f(bs), _f(hs), _f(bases), _f(acts),
                        _f(res))
        rc = self.lib.fluhip_pool_bufnmf_job_f32(self.h, ctypes.byref(job), _cb(progress), None)
        if rc not in (OK, CANCELLED):
            raise FluhipError(rc, self.lib.fluhip_pool_last_error(self.h).decode())
        return bases, acts, res, rc

    def bufnmf_ragged(self, audios, win, fft, hop, K, iters, seed=42, updateW=True, updateH=True, seeds=None, progress=None):
        """buffers of different lengths: lists of per-buffer bases [K,F] and activations [K,T_i]"""
        audios = [np.ascontiguousarray(a, dtype=np.float32) for a in audios]
        count = len(audios)
        F = fft // 2 + 1
        Ts = [int(self.lib.fluhip_stft_num_frames(a.shape[0], win, hop)) for a in audios]
        bases = [np.empty((K, F), dtype=np.float32) for _ in range(count)]
        acts = [np.empty((K, T), dtype=np.float32) for T in Ts]
        fpp = ctypes.POINTER(ctypes.c_float)
        ap = (fpp * count)(*[a.ctypes.data_as(fpp) for a in audios])
        bp = (fpp * count)(*[b.ctypes.data_as(fpp) for b in bases])
        cp = (fpp * count)(*[c.ctypes.data_as(fpp) for c in acts])
        lens = (ctypes.c_int64 * count)(*[a.shape[0] for a in audios])
        sarr = None if seeds is None else np.ascontiguousarray(seeds, dtype=np.int64)
        cb = _cb(progress)
        rc = self.lib.fluhip_pool_bufnmf_ragged_f32(self.h, ap, lens, count, win, fft, hop, K, iters, int(updateW), int(updateH),
                                                    seed, sarr.ctypes.data_as(_ip) if sarr is not None else None, bp, cp, cb, None)
        if rc not in (OK, CANCELLED):
            raise FluhipError(rc, self.lib.fluhip_pool_last_error(self.h).decode())
        return bases, acts, rc
