"""GPU parity tests: the HIP path, called through the C ABI (include/flucoma_hip.h), against
the CPU oracle on the same seeded inputs and against the committed golden fixtures.

Bars (BASELINE.json north_star): frame indexing bit-exact; spectrogram <= 1e-12 relative (f64
vs f64); W, H within 1e-5 relative -- the f64 kernels are expected to sit many orders below
that, which the *_tight assertions record.
"""
import os
import numpy as np
import pytest

from helpers import TOL_FACTORS, TOL_FACTORS_TIGHT, TOL_STFT, elementwise_rel_err, rel_err

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------------------------------
# K1: STFT
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("wfh", [(1024, 1024, 512), (2048, 2048, 512), (512, 1024, 256)])
def test_stft_golden(ctx, golden, wfh):
    win, fft, hop = wfh
    sig = golden["g2_signal"]
    spec, mag = ctx.stft(sig, win, fft, hop)
    key = f"g2_{win}_{fft}_{hop}"
    assert spec.shape[0] == int(golden[key + "_T"][0])          # frame count: exact
    rows = golden[key + "_rows"]
    scale = np.abs(golden[key + "_spec"]).max()
    assert np.abs(spec[rows] - golden[key + "_spec"]).max() / scale < TOL_STFT
    assert np.abs(mag[rows] - golden[key + "_mag"]).max() / scale < TOL_STFT
    assert abs(mag.sum() - golden[key + "_magsum"][0]) / golden[key + "_magsum"][0] < TOL_STFT
    assert np.all(spec[:, 0].imag == 0) and np.all(spec[:, -1].imag == 0)  # util/FFT.hpp:99-101


@pytest.mark.parametrize("n,win,fft,hop", [
    (44100, 1024, 1024, 512), (44100, 2048, 2048, 512), (30000, 4096, 4096, 1024),
    (10000, 1000, 1024, 250),   # non power-of-two window, zero-padded tail
    (5000, 64, 64, 16), (777, 16, 16, 4), (100, 8, 8, 2), (50, 4, 4, 1),
    (1, 1024, 1024, 512),       # a single sample: T = 1
    (511, 1024, 1024, 512), (512, 1024, 1024, 512), (513, 1024, 1024, 512),  # ragged around one hop
    (20000, 8192, 8192, 2048),  # largest in-LDS size (twiddles from global memory)
    (60000, 16384, 16384, 4096), (70000, 20000, 32768, 8192), (140000, 65536, 65536, 16384),  # global-memory passes
    (3000, 256, 512, 64), (3000, 32, 2048, 8),
    (12345, 1024, 1024, 333), (5001, 2048, 2048, 77), (30001, 4096, 4096, 999), (8191, 2048, 4096, 511),  # odd hops / lengths
])
def test_stft_vs_oracle(ctx, oracle, onp, n, win, fft, hop):
    x = onp.synth_audio(n, 1000 + n % 17)
    spec, mag = ctx.stft(x, win, fft, hop)
    rspec, rmag = oracle.stft_f32(x, win, fft, hop)
    assert spec.shape == rspec.shape == ((n + hop) // hop, fft // 2 + 1)
    scale = max(np.abs(rspec).max(), 1e-30)
    assert np.abs(spec - rspec).max() / scale < TOL_STFT
    assert np.abs(mag - rmag).max() / scale < TOL_STFT


def test_stft_f64_input_and_stride(ctx, oracle):
    rs = np.random.RandomState(4)
    x = rs.standard_normal(9000)
    spec, mag = ctx.stft(x, 512, 512, 128)
    rspec, rmag = oracle.stft(x, 512, 512, 128)
    assert rel_err(spec, rspec) < TOL_STFT and rel_err(mag, rmag) < TOL_STFT
    # strided view (BufferAdaptor::samps of an interleaved 2-channel buffer: cc/BufferAdaptor.hpp:61-66)
    inter = np.zeros(18000, dtype=np.float32)
    inter[0::2] = x.astype(np.float32)
    inter[1::2] = 99.0
    spec2, mag2 = ctx.stft(inter, 512, 512, 128, stride=2)
    rspec2, rmag2 = oracle.stft_f32(x.astype(np.float32), 512, 512, 128)
    assert rel_err(mag2, rmag2) < TOL_STFT


def test_stft_linearity_and_shift(ctx, onp):
    """size-independent properties at a BASELINE shape (fft 2048 / hop 512)"""
    n, win, fft, hop = 441000, 2048, 2048, 512
    a = onp.synth_audio(n, 1).astype(np.float64)
    b = onp.synth_audio(n, 2).astype(np.float64)
    sa, _ = ctx.stft(a, win, fft, hop)
    sb, _ = ctx.stft(b, win, fft, hop)
    sab, _ = ctx.stft(2.0 * a - 3.0 * b, win, fft, hop)
    assert sa.shape == (862, 1025)
    assert rel_err(sab, 2.0 * sa - 3.0 * sb) < 1e-11
    # shifting the input by one hop shifts the frames by one
    sh, _ = ctx.stft(np.concatenate([np.zeros(hop), a])[:n], win, fft, hop)
    assert rel_err(sh[3:-3], sa[2:-4]) < 1e-11
    # Parseval per frame on an interior frame: sum |x w|^2 = (|X0|^2 + 2 sum |Xk|^2 + |XN|^2)/fft
    w = onp.hann(win)
    t = 400
    fr = a[t * hop - win // 2: t * hop - win // 2 + win] * w
    e = (np.abs(sa[t, 0]) ** 2 + 2 * (np.abs(sa[t, 1:-1]) ** 2).sum() + np.abs(sa[t, -1]) ** 2) / fft
    assert abs(e - (fr * fr).sum()) / (fr * fr).sum() < 1e-12


def test_stft_rejects_bad_shapes(ctx):
    import fluhip
    x = np.zeros(1000, dtype=np.float32)
    with pytest.raises(fluhip.FluhipError):
        ctx.stft(x, 1024, 1000, 512)      # fft not a power of two
    with pytest.raises(fluhip.FluhipError):
        ctx.stft(x, 2048, 1024, 512)      # fft < win
    with pytest.raises(fluhip.FluhipError):
        ctx.stft(x, 1024, 131072, 512)    # beyond the reference's own limit of 65536 (util/FFT.hpp:113-122)


# ---------------------------------------------------------------------------------------
# NMF
# ---------------------------------------------------------------------------------------
def test_nmf_repeatable_with_seed(ctx):
    """tests/algorithms/public/TestNMF.cpp:11-46 verbatim, through the C ABI"""
    X = np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9.0]])
    a = ctx.nmf_process(X, 2, 1, True, True, 42)
    b = ctx.nmf_process(X, 2, 1, True, True, 42)
    c = ctx.nmf_process(X, 2, 1, True, True, 5063)
    d = ctx.nmf_process(X, 2, 1, True, True, 5063)
    for i in range(3):
        assert np.array_equal(a[i], b[i]) and np.array_equal(c[i], d[i])
        assert not np.array_equal(a[i], c[i])
        assert np.isfinite(a[i]).all()


@pytest.mark.parametrize("seed", [42, 5063])
@pytest.mark.parametrize("iters", [1, 50])
def test_nmf_tiny_golden(ctx, golden, seed, iters):
    W1, H1, V1, rc = ctx.nmf_process(golden["g4_X"], 2, iters, True, True, seed)
    assert rc == 0
    assert rel_err(W1, golden[f"g4_s{seed}_i{iters}_W"]) < TOL_FACTORS_TIGHT
    assert rel_err(H1, golden[f"g4_s{seed}_i{iters}_H"]) < TOL_FACTORS_TIGHT
    assert rel_err(V1, golden[f"g4_s{seed}_i{iters}_V"]) < TOL_FACTORS_TIGHT


@pytest.mark.parametrize("mode", ["u11", "u10", "u01", "u00", "seeded"])
def test_nmf_g5_golden(ctx, golden, mode):
    X, W0, H0 = golden["g5_X"], golden["g5_W0"], golden["g5_H0"]
    if mode == "seeded":
        uw, uh, iters, w0, h0 = True, True, 200, W0, H0
    else:
        uw, uh = mode[1] == "1", mode[2] == "1"
        iters = 200 if (uw or uh) else 0
        w0, h0 = (None if uw else W0), (None if uh else H0)
    W1, H1, V1, rc = ctx.nmf_process(X, 4, iters, uw, uh, 42, w0, h0)
    assert rc == 0
    for got, name in ((W1, "W"), (H1, "H"), (V1, "V")):
        e = rel_err(got, golden[f"g5_{mode}_{name}"])
        assert e < TOL_FACTORS, (mode, name, e)
        assert e < TOL_FACTORS_TIGHT, (mode, name, e)


@pytest.mark.parametrize("T,F,K,iters", [
    (87, 513, 3, 50),      # c1-like rank (padded to 16 inside)
    (200, 1025, 16, 30),   # c2 bins / rank
    (173, 1025, 32, 30),   # c4 bins / rank
    (97, 257, 48, 10), (64, 129, 64, 10), (50, 100, 100, 5), (40, 65, 128, 5),
    (33, 17, 1, 20), (16, 16, 16, 5), (17, 33, 5, 5), (1, 9, 2, 3), (9, 1, 2, 3),
])
def test_nmf_vs_oracle(ctx, oracle, T, F, K, iters):
    rs = np.random.RandomState(T * 7 + F)
    X = np.abs(rs.standard_normal((T, 3)) @ rs.standard_normal((3, F))) + 0.01 * rs.uniform(0, 1, (T, F))
    W1, H1, V1, rc = ctx.nmf_process(X, K, iters, True, True, 42)
    rW, rH, rV, _ = oracle.nmf_process(X, K, iters, True, True, 42)
    assert rc == 0
    assert rel_err(W1, rW) < TOL_FACTORS_TIGHT
    assert rel_err(H1, rH) < TOL_FACTORS_TIGHT
    assert rel_err(V1, rV) < TOL_FACTORS_TIGHT
    # column L2 norms of W are 1 after the last update (alg/NMF.hpp:162)
    assert np.allclose(np.sqrt((W1 * W1).sum(axis=1)), 1.0, atol=1e-12)


def test_nmf_strided_input(ctx, oracle):
    rs = np.random.RandomState(3)
    big = np.abs(rs.standard_normal((40, 80)))
    X = big[:, :33]  # ldx = 80
    W1, H1, V1, _ = ctx.nmf_process(X, 4, 10, True, True, 7)
    rW, rH, rV, _ = oracle.nmf_process(np.ascontiguousarray(X), 4, 10, True, True, 7)
    assert rel_err(W1, rW) < TOL_FACTORS_TIGHT and rel_err(H1, rH) < TOL_FACTORS_TIGHT


def test_nmf_two_stride_views(ctx, oracle):
    """fluhip_nmf_process_views_f64: every matrix a (rows, cols, rowStride, colStride) view like FluidTensorView /
    asEigen's Stride<Dynamic, Dynamic> (util/FluidEigenMappings.hpp:35-225) -- a transposed X (unit row stride), a
    sub-block with two non-unit strides, seeds and outputs living in transposed / padded arrays -- against the oracle
    on contiguous copies"""
    rs = np.random.RandomState(11)
    T, F, K, iters = 70, 45, 5, 9
    base = np.abs(rs.standard_normal((F, T)))                 # an F x T matrix; X is its transpose() view
    W0 = np.abs(rs.standard_normal((F, K))) + 0.1             # seeds stored the other way round as well
    H0 = np.abs(rs.standard_normal((K + 3, T + 2))) + 0.1
    W1 = np.zeros((F, K)); H1 = np.zeros((T, K + 4)); V1 = np.zeros((F + 5, T))
    rc = ctx.nmf_process_views(base.T, K, iters, seed=3, W0=W0.T, H0=H0[:K, :T].T, W1=W1.T, H1=H1[:, :K], V1=V1[:F].T)
    assert rc == 0
    rW, rH, rV, _ = oracle.nmf_process(np.ascontiguousarray(base.T), K, iters, True, True, 3,
                                       W0=np.ascontiguousarray(W0.T), H0=np.ascontiguousarray(H0[:K, :T].T))
    assert rel_err(W1.T, rW) < TOL_FACTORS_TIGHT and rel_err(H1[:, :K], rH) < TOL_FACTORS_TIGHT
    assert rel_err(V1[:F].T, rV) < TOL_FACTORS_TIGHT and not V1[F:].any() and not H1[:, K:].any()
    # both strides non-unit: every second row and column of a larger matrix; random init
    big = np.abs(rs.standard_normal((2 * T, 2 * F + 1)))
    Xs = big[::2, 1::2][:, :F]
    W1b = np.zeros((K, F)); H1b = np.zeros((T, K))
    assert ctx.nmf_process_views(Xs, K, iters, seed=7, W1=W1b, H1=H1b) == 0
    rW, rH, _, _ = oracle.nmf_process(np.ascontiguousarray(Xs), K, iters, True, True, 7)
    assert rel_err(W1b, rW) < TOL_FACTORS_TIGHT and rel_err(H1b, rH) < TOL_FACTORS_TIGHT
    # a ONE-ROW view with a non-unit column stride (a[:, ::2] of a 1 x 2F matrix) is not a contiguous row: X must be
    # gathered with its stride and V1 scattered with it, leaving the elements in between alone
    row = np.abs(rs.standard_normal((1, 2 * F))) + 0.01
    W1c = np.zeros((K, F)); H1c = np.zeros((1, K)); V1c = np.full((1, 2 * F), -1.0)
    assert ctx.nmf_process_views(row[:, ::2], K, iters, seed=7, W1=W1c, H1=H1c, V1=V1c[:, ::2]) == 0
    rW, rH, rV, _ = oracle.nmf_process(np.ascontiguousarray(row[:, ::2]), K, iters, True, True, 7)
    assert rel_err(W1c, rW) < TOL_FACTORS_TIGHT and rel_err(H1c, rH) < TOL_FACTORS_TIGHT
    assert rel_err(V1c[:, ::2], rV) < TOL_FACTORS_TIGHT and (V1c[:, 1::2] == -1.0).all()
    # and the mirror image: a one-COLUMN view (a single bin) with a non-unit row stride
    col = np.abs(rs.standard_normal((2 * T, 3))) + 0.01
    W1d = np.zeros((2, 1)); H1d = np.zeros((T, 2)); V1d = np.full((2 * T, 3), -1.0)
    assert ctx.nmf_process_views(col[::2, 1:2], 2, iters, seed=7, W1=W1d, H1=H1d, V1=V1d[::2, 1:2]) == 0
    rW, rH, rV, _ = oracle.nmf_process(np.ascontiguousarray(col[::2, 1:2]), 2, iters, True, True, 7)
    assert rel_err(W1d, rW) < TOL_FACTORS_TIGHT and rel_err(H1d, rH) < TOL_FACTORS_TIGHT
    assert rel_err(V1d[::2, 1:2], rV) < TOL_FACTORS_TIGHT and (V1d[1::2] == -1.0).all() and (V1d[:, 0] == -1.0).all()
    # shape checks of alg/NMF.hpp:109-110, 121-122
    import fluhip
    with pytest.raises(fluhip.FluhipError):
        ctx.nmf_process_views(Xs, K, 1, W0=np.ones((K + 1, F)))


def test_nmf_cancel_overshoot_is_bounded(ctx, oracle):
    """the callback refuses iteration 5: no later callback arrives, and the factors handed back are those of an
    iteration between 5 and 5 + 7 (include/flucoma_hip.h: the device is never more than 8 iterations ahead)"""
    import fluhip
    X = np.abs(np.random.RandomState(2).standard_normal((300, 129)))
    seen = []
    W1, H1, V1, rc = ctx.nmf_process(X, 6, 200, True, True, 1, progress=lambda it: seen.append(it) or it < 5)
    assert rc == fluhip.CANCELLED and seen == [1, 2, 3, 4, 5]
    errs = []
    for it in range(5, 13):
        rW, rH, _, _ = oracle.nmf_process(X, 6, it, True, True, 1)
        errs.append(max(rel_err(W1, rW), rel_err(H1, rH)))
    assert min(errs) < TOL_FACTORS_TIGHT, errs
    # fluhip_ctx_set_progress_lag(ctx, 1): the reference's exact behaviour (alg/NMF.hpp:175-176 returns AT the iteration whose
    # callback refuses) -- the factors handed back are those of iteration 5, no later one
    ctx.set_progress_lag(1)
    try:
        seen = []
        W1, H1, V1, rc = ctx.nmf_process(X, 6, 200, True, True, 1, progress=lambda it: seen.append(it) or it < 5)
        assert rc == fluhip.CANCELLED and seen == [1, 2, 3, 4, 5]
        rW, rH, _, _ = oracle.nmf_process(X, 6, 5, True, True, 1)
        assert rel_err(W1, rW) < TOL_FACTORS_TIGHT and rel_err(H1, rH) < TOL_FACTORS_TIGHT
        with pytest.raises(fluhip.FluhipError):
            ctx.set_progress_lag(0)
    finally:
        ctx.set_progress_lag(8)


def test_nmf_zero_columns_and_rows(ctx, oracle):
    """silent frames / empty bins: V has exact zeros (clamps at eps keep everything finite)"""
    rs = np.random.RandomState(8)
    X = np.abs(rs.standard_normal((48, 40)))
    X[5] = 0
    X[:, 7] = 0
    X[40:] = 0
    W1, H1, V1, _ = ctx.nmf_process(X, 3, 40, True, True, 42)
    rW, rH, rV, _ = oracle.nmf_process(X, 3, 40, True, True, 42)
    assert np.isfinite(W1).all() and np.isfinite(H1).all()
    assert rel_err(W1, rW) < TOL_FACTORS_TIGHT and rel_err(H1, rH) < TOL_FACTORS_TIGHT


def test_nmf_all_zero_input(ctx, oracle):
    """digital silence: V == 0 everywhere -- the factors collapse to zero after one iteration and stay finite (every
    denominator is clamped at eps, alg/NMF.hpp:158-170), on the frame-strip schedule (rank 3) and the split one (rank 32)"""
    X = np.zeros((70, 129))
    for K in (3, 32):
        W1, H1, V1, _ = ctx.nmf_process(X, K, 6, True, True, 42)
        rW, rH, rV, _ = oracle.nmf_process(X, K, 6, True, True, 42)
        assert np.isfinite(W1).all() and np.isfinite(H1).all() and np.isfinite(V1).all()
        assert np.array_equal(W1, rW) and np.array_equal(H1, rH)


def test_nmf_progress_and_cancel(ctx):
    import fluhip
    X = np.abs(np.random.RandomState(1).standard_normal((64, 33)))
    calls = []
    W1, H1, V1, rc = ctx.nmf_process(X, 4, 25, True, True, 1, progress=lambda it: calls.append(it) or True)
    assert rc == fluhip.OK and calls == list(range(1, 26))      # alg/NMF.hpp:175: cb(i + 1), every iteration
    calls2 = []

    def cancel_at_5(it):
        calls2.append(it)
        return it < 5

    W1, H1, V1, rc = ctx.nmf_process(X, 4, 25, True, True, 1, progress=cancel_at_5)
    assert rc == fluhip.CANCELLED and calls2 == [1, 2, 3, 4, 5]  # stops at once (:176)


def test_nmf_single_buffer_large_uses_split_path(ctx, oracle):
    """a single c2-shaped buffer (few column strips) exercises the split-R partial-sum path"""
    rs = np.random.RandomState(21)
    T, F, K = 700, 1025, 16
    X = np.abs(rs.standard_normal((T, 5)) @ rs.standard_normal((5, F))) + 0.001
    W1, H1, V1, _ = ctx.nmf_process(X, K, 20, True, True, 42, want_v=False)
    rW, rH, rV, _ = oracle.nmf_process(X, K, 20, True, True, 42)
    assert rel_err(W1, rW) < TOL_FACTORS_TIGHT and rel_err(H1, rH) < TOL_FACTORS_TIGHT
    # run-to-run reproducibility (fixed-order reductions): TestNMF.cpp:31-39
    W2, H2, _, _ = ctx.nmf_process(X, K, 20, True, True, 42, want_v=False)
    assert np.array_equal(W1, W2) and np.array_equal(H1, H2)


@pytest.mark.parametrize("T,F,K,iters", [(60, 65, 130, 6), (150, 129, 200, 5), (40, 257, 256, 4), (300, 33, 144, 8)])
def test_nmf_rank_above_128(ctx, oracle, T, F, K, iters):
    """ranks beyond the fused kernels' 128 take the un-fused path (kernels_nmf_wide.hip): same results as the oracle"""
    rs = np.random.RandomState(K)
    X = np.abs(rs.standard_normal((T, 12)) @ rs.standard_normal((12, F))) + 0.001
    for uw, uh in ((True, True), (True, False), (False, True)):
        W1, H1, V1, _ = ctx.nmf_process(X, K, iters, uw, uh, 42)
        rW, rH, rV, _ = oracle.nmf_process(X, K, iters, uw, uh, 42)
        assert rel_err(W1, rW) < TOL_FACTORS_TIGHT and rel_err(H1, rH) < TOL_FACTORS_TIGHT, (uw, uh)
        assert rel_err(V1, rV) < TOL_FACTORS_TIGHT


def test_bufnmf_rank_above_128_corpus(ctx, oracle, onp):
    import fluhip
    B, n, win, fft, hop, K, iters = 3, 9000, 256, 256, 64, 160, 5
    audio = np.stack([onp.synth_audio(n, 8000 + b) for b in range(B)])
    c = fluhip.Corpus(ctx, B, n, win, fft, hop, K)
    assert c.plan()["kernel"] == 0
    c.set_audio(audio); c.stft(); c.nmf(iters, seed=7)
    mag, W1, H1 = c.read_f64()
    c.close()
    for b in range(B):
        rW, rH, _, _ = oracle.nmf_process(mag[b], K, iters, True, True, 7)
        assert rel_err(W1[b], rW) < TOL_FACTORS_TIGHT and rel_err(H1[b], rH) < TOL_FACTORS_TIGHT


def test_nmf_long_factor_rank128(ctx, oracle):
    """c3's extremes at a size the oracle finishes in seconds: rank 128 (the widest kernel form) and a
    factor with tens of thousands of rows (the column-normalisation partials no longer fit in LDS)"""
    rs = np.random.RandomState(22)
    T, F, K = 26000, 129, 128
    X = np.abs(rs.standard_normal((T, 9)) @ rs.standard_normal((9, F))) + 0.001
    W1, H1, V1, _ = ctx.nmf_process(X, K, 3, True, True, 42)
    rW, rH, rV, _ = oracle.nmf_process(X, K, 3, True, True, 42)
    assert rel_err(W1, rW) < TOL_FACTORS_TIGHT and rel_err(H1, rH) < TOL_FACTORS_TIGHT
    assert rel_err(V1, rV) < TOL_FACTORS_TIGHT


# ---------------------------------------------------------------------------------------
# BufNMF channel + corpus
# ---------------------------------------------------------------------------------------
def test_bufnmf_channel_vs_oracle(ctx, oracle, onp):
    x = onp.synth_audio(44100, 1001)
    bases, acts, rc = ctx.bufnmf_channel(x, 1024, 1024, 512, 5, 50, 42)
    rb, ra = oracle.bufnmf_channel(x, 1024, 1024, 512, 5, 50, 42)
    assert rc == 0 and bases.shape == (5, 513) and acts.shape == (5, 87)
    assert rel_err(bases, rb) < TOL_FACTORS and rel_err(acts, ra) < TOL_FACTORS
    assert rel_err(bases, rb) < 1e-6 and rel_err(acts, ra) < 1e-6   # f32 outputs: 1 ulp-ish
    assert acts.max() == pytest.approx(1.0, abs=1e-6)                # H / max(H): nrt/NMFClient.hpp:289-298


def test_bufnmf_c1_shape_golden(ctx, onp, golden):
    """BASELINE config 1 shape end to end against the committed fixture"""
    import hashlib
    x = onp.drum_like(453932)
    sha = np.frombuffer(hashlib.sha256(x.tobytes()).digest(), dtype=np.uint8)
    if not np.array_equal(sha, golden["g6_input_sha256"]):
        pytest.skip("numpy RandomState stream differs from the one the fixture was minted with")
    bases, acts, rc = ctx.bufnmf_channel(x, 1024, 1024, 512, 3, 50, 42)
    assert bases.shape == (3, 513) and acts.shape == (3, 887)
    pb, pa = golden["g6_probe_bases_idx"], golden["g6_probe_acts_idx"]
    assert np.allclose(bases[pb[:, 0], pb[:, 1]], golden["g6_probe_bases"], rtol=1e-5, atol=1e-7)
    assert np.allclose(acts[pa[:, 0], pa[:, 1]], golden["g6_probe_acts"], rtol=1e-5, atol=1e-7)
    sums = golden["g6_sums"]
    assert abs(bases.astype(np.float64).sum() - sums[0]) / sums[0] < 1e-6
    assert abs(acts.astype(np.float64).sum() - sums[1]) / sums[1] < 1e-6


def test_c1_on_the_named_input(ctx, oracle):
    """BASELINE config 1 on the input it names: Nicol-LoopE-M.wav's own 453 932 samples (tests/golden/reference_c1.npz,
    tools/make_reference_c1_fixture.py), rank 3, fft 1024 / hop 512, 50 iterations, seed 42, through
    fluhip_bufnmf_channel_f32 -- T = 887, F = 513, the stored probes and sums of both oracles, and the whole float
    outputs against the C oracle run here on the same samples"""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_c1.npz"))
    win, fft, hop, K, iters, seed = (int(v) for v in g["params"])
    x = g["pcm16"].astype(np.float32) / 32768.0
    bases, acts, rc = ctx.bufnmf_channel(x, win, fft, hop, K, iters, seed)
    assert rc == 0 and bases.shape == (3, 513) and acts.shape == (3, 887)
    assert tuple(g["frames_bins"]) == (acts.shape[1], bases.shape[1])
    pb, pa = g["probe_bases_idx"], g["probe_acts_idx"]
    for tag in ("c", "np"):
        assert np.allclose(bases[pb[:, 0], pb[:, 1]], g["probe_bases_" + tag], rtol=1e-5, atol=1e-8)
        assert np.allclose(acts[pa[:, 0], pa[:, 1]], g["probe_acts_" + tag], rtol=1e-5, atol=1e-8)
        sums = g["sums_" + tag]
        assert abs(bases.astype(np.float64).sum() - sums[0]) / sums[0] < 1e-6
        assert abs(acts.astype(np.float64).sum() - sums[1]) / sums[1] < 1e-6
    rb, ra = oracle.bufnmf_channel(x, win, fft, hop, K, iters, seed)
    assert rel_err(bases, rb) < 1e-6 and rel_err(acts, ra) < 1e-6
    assert acts.max() == pytest.approx(1.0, abs=1e-6)
    # and the f64 factors of the same job through the corpus form (the strip schedule: rank <= 16, one buffer)
    import fluhip
    c = fluhip.Corpus(ctx, 1, len(x), win, fft, hop, K)
    assert (c.T, c.F) == (887, 513)
    c.set_audio(x[None, :]); c.stft(); c.nmf(iters, seed=seed)
    mag, W1, H1 = c.read_f64()
    c.close()
    _, rmag = oracle.stft_f32(x, win, fft, hop)
    rW, rH, _, _ = oracle.nmf_process(rmag, K, iters, True, True, seed)
    assert rel_err(mag[0], rmag) < TOL_STFT
    assert rel_err(W1[0], rW) < TOL_FACTORS_TIGHT and rel_err(H1[0], rH) < TOL_FACTORS_TIGHT
    assert elementwise_rel_err(W1[0], rW) < TOL_FACTORS and elementwise_rel_err(H1[0], rH) < TOL_FACTORS


def test_bufnmf_seeded_and_fixed_bases(ctx, oracle, onp):
    """basesMode Seed / Fixed (nrt/NMFClient.hpp:246-258, 268-271)"""
    x = onp.synth_audio(22050, 1002)
    win, fft, hop, K = 512, 512, 128, 4
    _, mag = oracle.stft_f32(x, win, fft, hop)
    rs = np.random.RandomState(0)
    seedW = rs.uniform(0.05, 1.0, (K, fft // 2 + 1)).astype(np.float32)
    # Fixed bases: only H updates
    bases, acts, rc = ctx.bufnmf_channel(x, win, fft, hop, K, 30, 42, updateW=False, bases_seed=seedW)
    rW, rH, _, _ = oracle.nmf_process(mag, K, 30, False, True, 42, W0=seedW.astype(np.float64))
    rb, ra = oracle.bufnmf_writeback(rW, rH)
    assert rel_err(acts, ra) < 1e-6 and rel_err(bases, rb) < 1e-6
    # Seeded bases: both update
    bases, acts, rc = ctx.bufnmf_channel(x, win, fft, hop, K, 30, 42, bases_seed=seedW)
    rW, rH, _, _ = oracle.nmf_process(mag, K, 30, True, True, 42, W0=seedW.astype(np.float64))
    rb, ra = oracle.bufnmf_writeback(rW, rH)
    assert rel_err(acts, ra) < 1e-6 and rel_err(bases, rb) < 1e-6


@pytest.mark.parametrize("T,F,K,iters,seed", [(50, 513, 5, 10, 42), (300, 1025, 32, 10, 5063), (7, 33, 3, 25, 1),
                                              (2000, 513, 16, 10, 42)])
def test_process_frames_vs_oracle(ctx, oracle, T, F, K, iters, seed):
    """NMF::processFrame (alg/NMF.hpp:45-89) on every frame of a magnitude matrix, the solve
    NMFMatch / NMFFilter run per spectral frame (10 iterations by default, rt/NMFMatchClient.hpp:116)"""
    rng = np.random.default_rng(T + K)
    W0 = rng.random((K, F)) ** 3
    acts = rng.random((T, K)) * (rng.random((T, K)) < 0.4)
    X = acts @ W0 + 1e-3 * rng.random((T, F))
    X[0, :7] = 0.0                                   # exercises the max(x, eps) clamp
    W0[0, :3] = 0.0                                  # and the max(W, eps) clamp
    H, V = ctx.nmf_process_frames(X, W0, iters, seed)
    rH, rV = oracle.nmf_process_frames(X[: min(T, 64)], W0, iters, seed)
    n = rH.shape[0]
    assert rel_err(H[:n], rH) < TOL_FACTORS_TIGHT and rel_err(V[:n], rV) < TOL_FACTORS_TIGHT
    assert np.isfinite(H).all() and (H >= 0).all()
    # frames are independent: a frame's activations do not depend on its neighbours
    H2, _ = ctx.nmf_process_frames(X[::-1].copy(), W0, iters, seed, want_v=False)
    assert rel_err(H2[::-1], H) < 1e-12


def _lowrank_spectrogram(T, F, r, seed):
    rs = np.random.RandomState(seed)
    scales = np.linspace(3.0, 0.3, r)            # well separated singular values: the vectors are well conditioned
    return (np.abs(rs.standard_normal((T, r))) * scales) @ np.abs(rs.standard_normal((r, F))) + 1e-3 * rs.uniform(0, 1, (T, F))


@pytest.mark.parametrize("T,F,amount,min_rank,max_rank", [(60, 33, 0.8, 0, 10), (200, 129, 0.5, 2, 16), (40, 65, 0.0, 3, 8),
                                                           (300, 513, 0.9, 1, 24)])
def test_nndsvd_method0_vs_oracle(ctx, onp, T, F, amount, min_rank, max_rank):
    """NNDSVD::process, method 0 (|U_k|, |S_k V_k^T|) -- independent of the SVD's sign convention, so
    rocSOLVER's factors must give what LAPACK's give; the rank rule (coverage of the singular-value sum) too"""
    X = _lowrank_spectrogram(T, F, 12, T + F)
    W, H, k = ctx.nndsvd(X, max_rank, min_rank, max_rank, amount, 0, 42)
    rW, rH, rk, U, s, VT = onp.nndsvd(X, max_rank, min_rank, max_rank, amount, 0, 42)
    assert k == rk and W.shape == rW.shape and H.shape == rH.shape
    assert rel_err(W, rW) < 1e-8 and rel_err(H, rH) < 1e-8
    assert (W[k:] == 0).all() and (H[:, k:] == 0).all()


@pytest.mark.parametrize("method", [1, 2, 3])
def test_nndsvd_split_methods(ctx, onp, method):
    """methods 1..3 follow the sign of each singular pair (as the reference follows Eigen's): every component must
    equal the oracle's construction for one of the two signs; zero fills as specified"""
    T, F, K = 120, 65, 8
    X = _lowrank_spectrogram(T, F, 10, 5)
    W, H, k = ctx.nndsvd(X, K, K, K, 0.0, method, 42)
    assert k == K
    _, s_, _ = np.linalg.svd(X.T, full_matrices=False)[0:3]
    U, s, VT = np.linalg.svd(X.T, full_matrices=False)
    eps, mean = 2.220446049250313e-16, float(X.mean())
    for j in range(K):
        best = np.inf
        for sign in (1.0, -1.0):
            Uj, VTj = U.copy(), VT.copy()
            Uj[:, j] *= sign; VTj[j] *= sign
            cW, cH, _ = onp.nndsvd_from_svd(Uj, s, VTj, X, K, K, K, 0.0, 3, 42)   # un-filled construction
            mw, mh = cW[j] >= eps, cH[:, j] >= eps
            err = max(np.abs(W[j][mw] - cW[j][mw]).max() / np.abs(cW[j]).max(),
                      np.abs(H[:, j][mh] - cH[:, j][mh]).max() / np.abs(cH[:, j]).max())
            if err < best:
                best, zw, zh = err, ~mw, ~mh
        assert best < 1e-8, (j, best)
        if method == 1:
            assert ((W[j][zw] >= eps) & (W[j][zw] <= mean * 0.001)).all() and ((H[:, j][zh] >= eps) & (H[:, j][zh] <= mean * 0.001)).all()
        elif method == 2:
            assert np.allclose(W[j][zw], mean, rtol=1e-12) and np.allclose(H[:, j][zh], mean, rtol=1e-12)
        else:
            assert (W[j][zw] < eps).all() and (H[:, j][zh] < eps).all()


def test_bufnmfseed_vs_oracle(ctx, oracle, onp):
    """BufNMFSeed (nrt/NMFSeedClient.hpp:73-131): STFT -> magnitude -> NNDSVD (method 0) -> float bases and
    activations scaled by 1 / max(H)"""
    x = onp.synth_audio(30000, 77)
    win, fft, hop, max_rank = 1024, 1024, 256, 12
    bases, acts, k = ctx.bufnmfseed(x, win, fft, hop, 1, max_rank, 0.7, 0, 42)
    _, mag = oracle.stft_f32(x, win, fft, hop)
    rW, rH, rk, *_ = onp.nndsvd(mag, max_rank, 1, max_rank, 0.7, 0, 42)
    assert k == rk and bases.shape == (max_rank, 513) and acts.shape == (max_rank, mag.shape[0])
    ra = (rH.T.astype(np.float32) * np.float32(1.0 / rH.max()))
    assert rel_err(bases[:k], rW[:k].astype(np.float32)) < 1e-5 and rel_err(acts[:k], ra[:k]) < 1e-5
    assert (bases[k:] == 0).all() and (acts[k:] == 0).all()
    assert abs(float(acts.max()) - 1.0) < 1e-6


def test_concurrent_contexts(onp):
    """the reference runs one std::thread per job (FluidNRTClientWrapper.hpp:1048): four host threads, each with
    its own context, must get bit-identical results to the serial run (shared block pool, per-context streams)"""
    import threading
    import fluhip
    jobs = [(onp.synth_audio(20000 + 3000 * i, 300 + i), 512, 512, 128, 3 + i, 12, 42) for i in range(4)]
    serial = []
    c0 = fluhip.Context(0)
    for j in jobs:
        serial.append(c0.bufnmf_channel(*j)[:2])
    c0.close()
    out, errs = [None] * 4, []

    def work(i):
        try:
            c = fluhip.Context(0)
            for _ in range(3):
                out[i] = c.bufnmf_channel(*jobs[i])[:2]
            c.close()
        except Exception as e:  # pragma: no cover
            errs.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for i in range(4):
        assert np.array_equal(out[i][0], serial[i][0]) and np.array_equal(out[i][1], serial[i][1])


def test_reference_testnmf_cases(ctx, oracle):
    """the two cases of the reference's own tests/algorithms/public/TestNMF.cpp, same inputs, through the HIP path:
    :11-46 (3x3 matrix, rank 2, 1 iteration, seeds 42 / 5063: equal for equal seeds, different otherwise) and
    :48-73 (processFrame, 0 iterations, seeds 42 / 7863) -- and equal to the oracle's values"""
    X = np.array([[1.0, 2, 3], [4, 5, 6], [7, 8, 9]])
    r = [ctx.nmf_process(X, 2, 1, True, True, sd)[:3] for sd in (42, 42, 5063, 5063)]
    for a, b in ((0, 1), (2, 3)):
        assert all(np.array_equal(r[a][i], r[b][i]) for i in range(3))
    assert all(not np.array_equal(r[1][i], r[2][i]) for i in range(3))
    oW, oH, oV, _ = oracle.nmf_process(X, 2, 1, True, True, 42)
    assert rel_err(r[0][0], oW) < TOL_FACTORS_TIGHT and rel_err(r[0][1], oH) < TOL_FACTORS_TIGHT
    assert rel_err(r[0][2], oV) < TOL_FACTORS_TIGHT
    x = np.array([[1.0, 0, 1, 0]])
    bases = np.array([[0.0, 0, 1, 0], [1, 0, 0, 0]])
    h = [ctx.nmf_process_frames(x, bases, 0, sd)[0] for sd in (42, 42, 7863)]
    assert np.array_equal(h[0], h[1]) and not np.array_equal(h[1], h[2])
    assert np.array_equal(h[0], oracle.nmf_process_frames(x, bases, 0, 42)[0])


def test_large_fft_roundtrip_and_bufnmf(ctx, oracle, onp):
    """fft sizes above the LDS-resident 8192 (the reference's shared FFT setup goes to 65536, util/FFT.hpp:113-122):
    BufSTFT forward / inverse round trip and a BufNMF channel with resynthesis, against the oracle"""
    x = onp.synth_audio(120000, 31)
    win = fft = 16384
    hop = 4096
    mag, ph = ctx.bufstft_forward(x, win, fft, hop, 1)
    rmag, rph = oracle.bufstft_forward(x, win, fft, hop, 1)
    assert rel_err(mag, rmag) < 1e-5
    y = ctx.bufstft_inverse(mag, ph, win, fft, hop, 1)
    assert np.abs(y[:x.size] - x).max() < 1e-4
    bases, acts, res, rc = ctx.bufnmf_channel(x, win, fft, hop, 3, 10, 42, resynth=True)
    rb, ra = oracle.bufnmf_channel(x, win, fft, hop, 3, 10, 42)
    assert rel_err(bases, rb) < 1e-6 and rel_err(acts, ra) < 1e-6
    assert rel_err(res.sum(axis=0), x) < 1e-4


def test_next_row_entry_points_reject_bad_arguments(ctx):
    """error convention of the boundary (Result::Status codes + message, no exceptions across the ABI) on the
    entry points of the "next" rows"""
    import fluhip
    X = np.abs(np.random.RandomState(0).standard_normal((20, 17)))
    W0 = np.abs(np.random.RandomState(1).standard_normal((3, 17)))
    with pytest.raises(fluhip.FluhipError):
        ctx.nmf_process_frames(X, W0, -1, 42)                      # negative iteration count
    with pytest.raises(AssertionError):
        ctx.nmf_process_frames(X, W0[:, :5], 3, 42)                # dictionary of another bin count (binding check)
    with pytest.raises(fluhip.FluhipError):
        ctx.nndsvd(X, 4, 0, 4, 0.0, 0, 42)                         # neither coverage nor minimum rank (NNDSVD.hpp:40)
    with pytest.raises(fluhip.FluhipError):
        ctx.nndsvd(X, 4, 0, 4, 0.5, 7, 42)                         # unknown method
    with pytest.raises(fluhip.FluhipError):
        ctx.nndsvd(X, 2, 6, 8, 0.0, 0, 42)                         # rank beyond the rows of W
    with pytest.raises(fluhip.FluhipError):
        ctx.bufnmfseed(np.zeros(0, dtype=np.float32), 1024, 1024, 512)  # no frames
    with pytest.raises(fluhip.FluhipError) as e:
        ctx.bufnmf_channel(np.zeros(1000, dtype=np.float32), 1024, 1000, 512, 3, 5, 42)
    assert "power of two" in e.value.message


def test_process_frames_golden(ctx):
    """the HIP path against the committed G7 vectors directly (no oracle in the loop)"""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_frames_v1.npz"))
    for seed, iters in ((42, 10), (5063, 10), (42, 0), (7, 100)):
        H, V = ctx.nmf_process_frames(g["g7_X"], g["g7_W0"], iters, seed)
        assert rel_err(H, g[f"g7_s{seed}_i{iters}_H"]) < TOL_FACTORS_TIGHT
        assert rel_err(V, g[f"g7_s{seed}_i{iters}_V"]) < TOL_FACTORS_TIGHT


def test_corpus_matches_per_buffer_oracle(ctx, oracle, onp):
    """the batched (corpus) form against independent per-buffer oracle runs, c4's fft/rank"""
    import fluhip
    B, n, win, fft, hop, K, iters = 12, 22050, 2048, 2048, 512, 32, 25
    audio = np.stack([onp.synth_audio(n, 1000 + b) for b in range(B)])
    c = fluhip.Corpus(ctx, B, n, win, fft, hop, K)
    assert (c.T, c.F) == ((n + hop) // hop, 1025)
    c.set_audio(audio)
    c.stft()
    c.nmf(iters, seed=42)
    mag, W1, H1 = c.read_f64()
    bases, acts = c.writeback()
    for b in (0, 5, 11):
        _, rmag = oracle.stft_f32(audio[b], win, fft, hop)
        assert rel_err(mag[b], rmag) < TOL_STFT
        rW, rH, _, _ = oracle.nmf_process(rmag, K, iters, True, True, 42)
        assert rel_err(W1[b], rW) < TOL_FACTORS_TIGHT and rel_err(H1[b], rH) < TOL_FACTORS_TIGHT
        rb, ra = oracle.bufnmf_writeback(rW, rH)
        assert rel_err(bases[b], rb) < 1e-6 and rel_err(acts[b], ra) < 1e-6
    c.close()


@pytest.mark.parametrize("B,n,win,fft,hop,K", [(5, 22051, 2048, 2048, 333, 6),     # odd length (odd buffer bases), odd hop
                                                (7, 9999, 1000, 1024, 250, 4),      # window shorter than the transform
                                                (3, 40001, 4096, 4096, 1024, 8),    # config 3's transform, partial last block
                                                (9, 3001, 1024, 1024, 512, 3),      # fewer frames than a block holds
                                                (2, 100, 1024, 1024, 512, 2)])      # a single frame per buffer
def test_corpus_block_stft_both_layouts(ctx, oracle, onp, B, n, win, fft, hop, K):
    """The block STFT kernel writes V frame-major AND bin-major in one pass: the frame-major copy against the oracle's
    spectrogram, the bin-major one through the H update that streams it (factors against the oracle), on shapes that
    exercise its edges -- unaligned buffers, ragged last blocks, clamped gathers on every frame"""
    import fluhip
    audio = np.stack([onp.synth_audio(n, 4000 + b) for b in range(B)])
    c = fluhip.Corpus(ctx, B, n, win, fft, hop, K)
    c.set_audio(audio); c.stft(); c.nmf(6, seed=42)
    mag, W1, H1 = c.read_f64()
    c.close()
    for b in sorted({0, B // 2, B - 1}):
        _, rmag = oracle.stft_f32(audio[b], win, fft, hop)
        assert mag[b].shape == rmag.shape
        assert rel_err(mag[b], rmag) < TOL_STFT
        rW, rH, _, _ = oracle.nmf_process(rmag, K, 6, True, True, 42)
        assert rel_err(W1[b], rW) < TOL_FACTORS_TIGHT and rel_err(H1[b], rH) < TOL_FACTORS_TIGHT


def test_corpus_fast_path_partial_updates(ctx, oracle, onp):
    """c4's per-buffer shape class (fft 2048, rank 32) in the batched regime: the plan must be the two-launch
    fast path (deferred column normalisation of W, Nyquist bin as a side column), and W-only / H-only runs --
    consecutive W updates read back their own un-normalised W' -- must match the oracle like full runs do."""
    import fluhip
    B, n, win, fft, hop, K, iters = 128, 220500, 2048, 2048, 512, 32, 12
    distinct = [onp.synth_audio(n, 4000 + b) for b in range(16)]
    audio = np.stack([distinct[b % 16] for b in range(B)])
    c = fluhip.Corpus(ctx, B, n, win, fft, hop, K)
    plan = c.plan()
    assert plan["kernel"] == 5 and plan["split_w"] == 1 and plan["split_h"] == 1
    assert plan["deferred_norm"] == 1 and plan["side_column"] == 1, plan
    c.set_audio(audio); c.stft()
    mag = c.read_f64(factors=False)[0]
    for uw, uh in ((True, False), (False, True), (True, True)):
        c.nmf(iters, seed=5063, updateW=uw, updateH=uh)
        _, W1, H1 = c.read_f64(mag=False)
        for b in (0, 7, 127):
            rW, rH, _, _ = oracle.nmf_process(mag[b], K, iters, uw, uh, 5063)
            assert rel_err(W1[b], rW) < TOL_FACTORS_TIGHT and rel_err(H1[b], rH) < TOL_FACTORS_TIGHT, (uw, uh, b)
    # a cancelled run still hands back a normalised dictionary (the deferred form never leaks)
    # (how many iterations ran before the cancel took effect is not observable in the reference either:
    # clients/nrt/NMFClient.hpp:273-274 returns kCancelled without writing any output)
    c.nmf(50, seed=42, progress=lambda it: it < 5)
    _, Wc, Hc = c.read_f64(mag=False)
    assert np.isfinite(Wc).all() and np.isfinite(Hc).all()
    assert np.allclose(np.sqrt((Wc * Wc).sum(axis=2)), 1.0, atol=1e-12)
    c.close()


def test_corpus_resynthesis_matches_single_channel_path(ctx, onp):
    """the corpus form of the resynthesis (third BufNMF output) against the single-channel entry point, which is
    itself checked against the oracle (test_resynthesis_vs_oracle)"""
    import fluhip
    B, n, win, fft, hop, K, iters = 3, 12000, 512, 512, 128, 4, 15
    audio = np.stack([onp.synth_audio(n, 5000 + b) for b in range(B)])
    c = fluhip.Corpus(ctx, B, n, win, fft, hop, K)
    with pytest.raises(fluhip.FluhipError):
        c.set_audio(audio); c.stft(); c.nmf(2, seed=42); c.resynth()       # spectrum not kept
    c.keep_spectrum(True)
    c.set_audio(audio); c.stft(); c.nmf(iters, seed=42)
    out = c.resynth()
    c.close()
    assert out.shape == (B, K, n)
    for b in range(B):
        _, _, res, _ = ctx.bufnmf_channel(audio[b], win, fft, hop, K, iters, 42, resynth=True)
        assert rel_err(out[b], res) < 1e-6
    # the components sum back to the input where the mask is a partition of unity (exponent 1 ratio masks)
    assert rel_err(out.sum(axis=1), audio) < 1e-4


@pytest.mark.parametrize("fft,hop", [(2048, 256), (2048, 512), (2048, 1024), (1024, 128), (1024, 256), (1024, 512)])
@pytest.mark.parametrize("B,n,K", [(3, 30001, 5), (2, 9000, 9), (1, 1500, 2)])
def test_corpus_batched_resynthesis(ctx, onp, B, n, K, fft, hop):
    """fft 2048 at hops of 256 / 512 / 1024 and fft 1024 at 128 / 256 / 512: the batched resynthesis (resynth_seq_kernel: every component of every buffer
    in one launch, the overlap-add in registers) against the single-channel entry point's frame + overlap-add kernels,
    themselves checked against the oracle; odd lengths, ranks that do not fill a workgroup of eight components, a buffer
    shorter than the window"""
    import fluhip
    win = fft
    iters = 6
    audio = np.stack([onp.synth_audio(n, 5200 + b) for b in range(B)])
    c = fluhip.Corpus(ctx, B, n, win, fft, hop, K)
    c.keep_spectrum(True)
    c.set_audio(audio); c.stft(); c.nmf(iters, seed=42)
    out = c.resynth()
    c.close()
    assert out.shape == (B, K, n) and np.isfinite(out).all()
    for b in range(B):
        _, _, res, _ = ctx.bufnmf_channel(audio[b], win, fft, hop, K, iters, 42, resynth=True)
        assert rel_err(out[b], res) < 1e-6, (b, rel_err(out[b], res))
    assert rel_err(out.sum(axis=1), audio) < 1e-4


def test_corpus_resynthesis_interleaved_and_pinned_copy(ctx, onp):
    """fluhip_corpus_resynth_interleaved_host: the resynthesis as an interleaved host buffer holds it (frames x channels),
    transposed on the device -- the same floats as the channel-major form, with and without padding columns; a corpus large
    enough (80 MB) that both results leave through the pinned staging blocks"""
    import fluhip
    B, n, win, fft, hop, K = 5, 500001, 1024, 1024, 512, 8
    audio = np.stack([onp.synth_audio(n, 5300 + b) for b in range(B)])
    c = fluhip.Corpus(ctx, B, n, win, fft, hop, K)
    c.keep_spectrum(True)
    c.set_audio(audio); c.stft(); c.nmf(3, seed=42)
    ref = c.resynth()                                        # [B][K][n]
    inter = c.resynth_interleaved()                          # [n][B K]
    wide = c.resynth_interleaved(B * K + 3)                  # [n][B K + 3], the last three columns untouched
    c.close()
    assert np.isfinite(ref).all() and np.abs(ref).max() > 0
    assert np.array_equal(inter, ref.reshape(B * K, n).T)
    assert np.array_equal(wide[:, :B * K], inter) and (wide[:, B * K:] == 0).all()


@pytest.mark.parametrize("n", [1, 100, 511, 512, 513, 2047, 2049])
def test_batched_resynthesis_of_tiny_buffers(ctx, oracle, onp, n):
    """buffers shorter than a hop / a window (one to five frames, all of them edge frames whose normaliser is the direct
    sum): the batched kernel against the oracle"""
    win = fft = 2048
    hop, K, iters = 512, 2, 3
    x = onp.synth_audio(max(n, 64), 31 + n)[:n]
    bases, acts, res, rc = ctx.bufnmf_channel(x, win, fft, hop, K, iters, 42, resynth=True)
    assert rc == 0 and res.shape == (K, n)
    spec, mag = oracle.stft_f32(x, win, fft, hop)
    W1, H1, V1, _ = oracle.nmf_process(mag, K, iters, True, True, 42)
    for k in range(K):
        ref = oracle.resynth_component(spec, W1, H1, V1, k, win, fft, hop, n)
        assert np.abs(res[k] - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1e-12) + 1e-9, (k, np.abs(res[k] - ref).max())


def test_resynthesis_at_bench_shape_adds_up(ctx, oracle, onp):
    """size-independent property at the bench workload's buffer shape (16 x 10 s, rank 32): exponent-1 ratio masks are a
    partition of unity, so the 32 components of a buffer add back up to its samples; and the interleaved form holds the
    same floats"""
    import fluhip
    B, n, K = 16, 441000, 32
    base = [onp.synth_audio(n, 1000 + b) for b in range(4)]
    audio = np.stack([base[b % 4] for b in range(B)])
    c = fluhip.Corpus(ctx, B, n, 2048, 2048, 512, K)
    c.keep_spectrum(True)
    c.set_audio(audio); c.stft(); c.nmf(5, seed=42)
    out = c.resynth()
    inter = c.resynth_interleaved()
    c.close()
    assert np.isfinite(out).all()
    assert np.abs(out.sum(axis=1) - audio).max() < 2e-4
    assert np.array_equal(inter, out.reshape(B * K, n).T)
    assert np.array_equal(out[0], out[4]) and not np.array_equal(out[0], out[1])      # equal buffers, equal results
    # and two components of one buffer against the oracle at full length (runs of 128 hop slots, rows shared through the LDS)
    spec, mag = oracle.stft_f32(audio[1], 2048, 2048, 512)
    W1, H1, V1, _ = oracle.nmf_process(mag, K, 5, True, True, 42)
    for k in (5, 31):
        ref = oracle.resynth_component(spec, W1, H1, V1, k, 2048, 2048, 512, n)
        assert np.abs(out[1, k] - ref).max() <= 1e-5 * np.abs(ref).max(), k


def test_ragged_corpus_batched_resynthesis(ctx, onp):
    """the same on buffers of different lengths (per-buffer frame counts and sample counts inside one launch)"""
    import fluhip
    lens = [30000, 4100, 52001, 700, 2048]
    win, fft, hop, K, iters = 2048, 2048, 512, 6, 5
    audios = [onp.synth_audio(n, 9400 + i) for i, n in enumerate(lens)]
    c = fluhip.RaggedCorpus(ctx, lens, win, fft, hop, K)
    c.keep_spectrum(True)
    c.set_audio(audios); c.stft(); c.nmf(iters, seed=42)
    res = c.resynth()
    c.close()
    for b, n in enumerate(lens):
        _, _, rr, rc = ctx.bufnmf_channel(audios[b], win, fft, hop, K, iters, 42, resynth=True)
        assert rc == 0 and res[b].shape == (K, n) and rel_err(res[b], rr) < 1e-6, (b, rel_err(res[b], rr))


@pytest.mark.parametrize("iters", [6, 200])
@pytest.mark.parametrize("K", [64, 128])
def test_corpus_side_column_wide_ranks(ctx, oracle, onp, K, iters):
    """ranks 64 and 128 in the batched regime: the Nyquist side column saves a whole pass of wavefronts there (65 column
    groups at 4 / 2 per wavefront: 17 / 33 strips per buffer -> 16 / 32), so the plan must take it, and match the oracle.
    These ranks run the pipeline form that refills one operand set in place (kernels_nmf5.hip MODE 2), whose round-3 wrong
    result at rank 32 was a timing-dependent fault in a FEW buffers of a full chip: so EVERY one of the 128 buffers is
    held bit-for-bit against the first replica of its input (4 distinct inputs, one seed -- the reference's own
    determinism bar, tests/algorithms/public/TestNMF.cpp:31-39), the 4 against the oracle, at 6 and at 200 iterations."""
    import fluhip
    B, n, win, fft, hop = 128, 60000, 2048, 2048, 512
    distinct = [onp.synth_audio(n, 6000 + b) for b in range(4)]
    audio = np.stack([distinct[b % 4] for b in range(B)])
    c = fluhip.Corpus(ctx, B, n, win, fft, hop, K)
    plan = c.plan()
    assert plan["kernel"] == 5 and plan["deferred_norm"] == 1 and plan["side_column"] == 1, plan
    c.set_audio(audio); c.stft(); c.nmf(iters, seed=42)
    mag, W1, H1 = c.read_f64()
    c.close()
    for b in range(4, B):
        assert np.array_equal(W1[b], W1[b % 4]) and np.array_equal(H1[b], H1[b % 4]), b
    for b in range(4):
        rW, rH, _, _ = oracle.nmf_process(mag[b], K, iters, True, True, 42)
        assert rel_err(W1[b], rW) < TOL_FACTORS_TIGHT and rel_err(H1[b], rH) < TOL_FACTORS_TIGHT
        assert elementwise_rel_err(W1[b], rW) < TOL_FACTORS and elementwise_rel_err(H1[b], rH) < TOL_FACTORS


@pytest.mark.parametrize("B,fft,T,takes", [
    (128, 2048, 730, 7), (128, 2048, 862, 7), (128, 2048, 1000, 3), (128, 2048, 1100, 1),
    (256, 1024, 200, 7), (256, 1024, 330, 7), (256, 1024, 440, 7), (256, 1024, 500, 3), (512, 512, 130, 7), (512, 512, 60, 7)])
def test_two_launch_iteration_at_every_strip_width(ctx, oracle, onp, B, fft, T, takes):
    """Round 6: at rank 32 the two-launch iteration keeps no column-sum accumulators in its loops -- the W update sums the
    denominators of the side-column partials the H update in front left, the H update sums the column partials the W update
    left (kernels_nmf5.hip DS = 2) -- and forms its first product with VGPR results (QV).  The forms exist per strip width:
    W update 8 groups per strip at every one-round shape (128 x 1025 bins, 256 x 513, 512 x 257), H update 2 .. 7 groups with
    the new form, 8 with the norm form that keeps its accumulators (the W update still takes the new one), 9 with the
    side-column form only.  `takes` = what the planner says the H update takes over (bit 2: column sums from the W update).
    Seven iterations (the first has no side partials to start from, the last H update leaves none): EVERY buffer bit for bit
    against the first replica of its input, three buffers against the oracle."""
    import ctypes
    import fluhip
    win, hop, K, iters = fft, fft // 4, 32, 7
    F = fft // 2 + 1
    n = (T - 1) * hop + 3
    out = (ctypes.c_int64 * 32)()
    assert ctx.lib.fluhip_debug_plan_shape(B, T, F, K, out) == 0
    assert (out[0], out[1], out[2], out[4], out[9], out[22]) == (5, 1, 1, 1, 0, takes), list(out)[:27]
    distinct = [onp.synth_audio(n, 7300 + b) for b in range(3)]
    audio = np.stack([distinct[b % 3] for b in range(B)])
    c = fluhip.Corpus(ctx, B, n, win, fft, hop, K)
    assert c.T == T and c.F == F
    c.set_audio(audio); c.stft(); c.nmf(iters, seed=42)
    mag0 = c.read_f64(factors=False)[0][:3].copy()
    _, W1, H1 = c.read_f64(mag=False)
    # W-only and H-only calls behind it cross the forms' entry and exit conditions (no side partials / no column partials)
    c.nmf(2, seed=42, updateH=False)
    _, W2, _ = c.read_f64(mag=False)
    c.close()
    for b in range(3, B):
        assert np.array_equal(W1[b], W1[b % 3]) and np.array_equal(H1[b], H1[b % 3]), b
    for b in range(3):
        rW, rH, _, _ = oracle.nmf_process(mag0[b], K, iters, True, True, 42)
        assert rel_err(W1[b], rW) < TOL_FACTORS_TIGHT and rel_err(H1[b], rH) < TOL_FACTORS_TIGHT, b
        assert elementwise_rel_err(W1[b], rW) < TOL_FACTORS and elementwise_rel_err(H1[b], rH) < TOL_FACTORS
    rW, _, _, _ = oracle.nmf_process(mag0[0], K, 2, True, False, 42)
    assert rel_err(W2[0], rW) < TOL_FACTORS_TIGHT


def test_h_update_of_more_than_64_strips_keeps_the_side_column_launch(ctx, oracle, onp):
    """long buffers at rank 64: the H update takes 80 strips per buffer, more than the 64 slices per buffer the side-column
    partials of its epilogue have room for (kernels_nmf.hip wnorm_side_part) -- the launcher must leave such shapes on the
    side-column launch (kernels_nmf5.hip launch5_ng `w <= kSideFromHSlots`; before round 5 the epilogue wrote past the
    area: wrong Nyquist rows of W from 65 strips, other pool blocks overwritten well above 128).  Every buffer against the
    first replica of its input, two against the oracle; FLUHIP_CANARY-style damage would also show in the second corpus."""
    import fluhip
    B, T, win, fft, hop, K, iters = 64, 5000, 2048, 2048, 512, 64, 4
    n = (T - 1) * hop
    assert ctx.lib.fluhip_debug_plan_kind(B, T, 1025, K) == 0 and ctx.lib.fluhip_debug_plan_h_update(B, T, 1025, K) == 0
    distinct = [onp.synth_audio(n, 7100 + b) for b in range(2)]
    audio = np.stack([distinct[b % 2] for b in range(B)])
    c = fluhip.Corpus(ctx, B, n, win, fft, hop, K)
    assert c.T == T and c.plan()["side_column"] == 1, (c.T, c.plan())
    c.set_audio(audio); c.stft(); c.nmf(iters, seed=42)
    mag0 = c.read_f64(factors=False)[0][:2].copy()
    _, W1, H1 = c.read_f64(mag=False)
    c.close()
    for b in range(2, B):
        assert np.array_equal(W1[b], W1[b % 2]) and np.array_equal(H1[b], H1[b % 2]), b
    for b in range(2):
        rW, rH, _, _ = oracle.nmf_process(mag0[b], K, iters, True, True, 42)
        assert rel_err(W1[b], rW) < TOL_FACTORS_TIGHT and rel_err(H1[b], rH) < TOL_FACTORS_TIGHT, b


def _offsize_compute_rank(K):
    """smallest form that holds K (kernels_nmf5.hip nmf_update5_compute_rank)"""
    for kc in (16, 24, 32, 40, 48, 56, 64, 72, 80, 88, 96, 104, 112, 128):
        if K <= kc:
            return kc


@pytest.mark.parametrize("K", [17, 20, 24, 25, 33, 40, 41, 48, 49, 56, 57, 65, 72, 73, 80, 88, 96, 100, 104, 105, 112, 113])
def test_offsize_ranks_compute_fewer_products(ctx, oracle, onp, K):
    """ranks between two array ranks (clients/nrt/NMFClient.hpp:68: `components` is any integer >= 1): the arrays keep rank 32 / 64 /
    128, the factor updates compute 6 | 10 / 12 / 14 | 18 .. 28 MFMAs per product (kernels_nmf5.hip KPM; round 5) -- a full
    chip of 128 buffers on the plain schedule (every replica bit for bit, two against the oracle), a stereo pair and a single
    buffer on the split schedule; ranks 25, 57, 113 stay on the padded forms"""
    import fluhip
    n, win, fft, hop, iters = 60000, 2048, 2048, 512, 6
    want = _offsize_compute_rank(K)
    distinct = [onp.synth_audio(n, 6100 + b) for b in range(4)]
    for B in (128, 2, 1):
        audio = np.stack([distinct[b % 4] for b in range(B)])
        c = fluhip.Corpus(ctx, B, n, win, fft, hop, K)
        plan = c.plan()
        c.set_audio(audio); c.stft(); c.nmf(iters, seed=42)
        mag, W1, H1 = c.read_f64()
        c.close()
        if B == 128:
            assert plan["compute_rank"] == want and plan["padded_rank"] == (32 if K <= 32 else (64 if K <= 64 else 128)), plan
        for b in range(4, B):
            assert np.array_equal(W1[b], W1[b % 4]) and np.array_equal(H1[b], H1[b % 4]), (B, b)
        for b in range(min(B, 2)):
            rW, rH, _, _ = oracle.nmf_process(mag[b], K, iters, True, True, 42)
            assert rel_err(W1[b], rW) < TOL_FACTORS_TIGHT and rel_err(H1[b], rH) < TOL_FACTORS_TIGHT, (B, b, plan)


def test_corpus_buffers_are_independent_and_order_free(ctx, onp):
    """sharding property: a buffer's result does not depend on which batch (or rank) holds it"""
    import fluhip
    B, n, win, fft, hop, K, iters = 9, 11025, 1024, 1024, 256, 16, 15
    audio = np.stack([onp.synth_audio(n, 2000 + b) for b in range(B)])
    c = fluhip.Corpus(ctx, B, n, win, fft, hop, K)
    c.set_audio(audio); c.stft(); c.nmf(iters, seed=42)
    _, W_all, H_all = c.read_f64()
    c.close()
    perm = np.array([4, 0, 8])
    c2 = fluhip.Corpus(ctx, 3, n, win, fft, hop, K)
    c2.set_audio(audio[perm]); c2.stft(); c2.nmf(iters, seed=42)
    _, W_sub, H_sub = c2.read_f64()
    c2.close()
    assert np.array_equal(W_sub, W_all[perm]) and np.array_equal(H_sub, H_all[perm])


def test_corpus_of_several_rounds_runs_round_major(ctx, oracle, onp):
    """a corpus of more buffers than one round of wavefronts holds (no progress callback) runs all iterations of a round's
    buffers before the next round's (corpus_iterate_loop: windows of the per-buffer arrays): buffers at the window
    edges against the oracle, partial updates, and a progress callback (iteration-major order) giving the same factors"""
    import fluhip
    B, n, win, fft, hop, K, iters = 2100, 2000, 256, 256, 64, 8, 7      # one strip per buffer: rounds of 1024, 1024, 52
    distinct = [onp.synth_audio(n, 8800 + b) for b in range(5)]
    audio = np.stack([distinct[b % 5] for b in range(B)])
    c = fluhip.Corpus(ctx, B, n, win, fft, hop, K)
    plan = c.plan()
    assert plan["split_w"] == 1 and plan["split_h"] == 1, plan
    c.set_audio(audio); c.stft()
    mag = c.read_f64(factors=False)[0]
    for uw, uh in ((True, True), (True, False), (False, True)):
        c.nmf(iters, seed=42, updateW=uw, updateH=uh)
        _, W1, H1 = c.read_f64(mag=False)
        for b in (0, 1023, 1024, 2047, 2048, 2099):
            rW, rH, _, _ = oracle.nmf_process(mag[b], K, iters, uw, uh, 42)
            assert rel_err(W1[b], rW) < TOL_FACTORS_TIGHT and rel_err(H1[b], rH) < TOL_FACTORS_TIGHT, (uw, uh, b)
    seen = []
    c.nmf(iters, seed=42, progress=lambda i: seen.append(i) or True)
    _, W2, H2 = c.read_f64(mag=False)
    c.nmf(iters, seed=42)
    _, W1, H1 = c.read_f64(mag=False)
    assert seen == list(range(1, iters + 1))
    assert np.array_equal(W1, W2) and np.array_equal(H1, H2)     # the order of independent jobs changes nothing
    c.close()


def test_corpus_per_buffer_seeds(ctx, oracle, onp):
    import fluhip
    B, n, win, fft, hop, K, iters = 3, 8000, 512, 512, 128, 4, 10
    audio = np.stack([onp.synth_audio(n, 3000 + b) for b in range(B)])
    c = fluhip.Corpus(ctx, B, n, win, fft, hop, K)
    c.set_audio(audio); c.stft(); c.nmf(iters, seeds=[42, 5063, 42])
    mag, W1, H1 = c.read_f64()
    for b, seed in enumerate((42, 5063, 42)):
        rW, rH, _, _ = oracle.nmf_process(mag[b], K, iters, True, True, seed)
        assert rel_err(W1[b], rW) < TOL_FACTORS_TIGHT and rel_err(H1[b], rH) < TOL_FACTORS_TIGHT
    c.close()


def _check_ragged(ctx, oracle, onp, lens, win, fft, hop, K, iters, check, seeds=None, seedW=None, uw=True):
    import fluhip
    audios = [onp.synth_audio(n, 9000 + i) for i, n in enumerate(lens)]
    c = fluhip.RaggedCorpus(ctx, lens, win, fft, hop, K)
    assert c.Ts == [(n + hop) // hop for n in lens] and c.T == max(c.Ts)
    c.set_audio(audios); c.stft()
    if seedW is not None:
        c.set_factors(seedW, None)
    c.nmf(iters, seed=42, seeds=seeds, updateW=uw)
    mag, W1, H1 = c.read_f64()
    bases, acts = c.writeback()
    plan = c.plan()
    c.close()
    for b in check:
        T = (lens[b] + hop) // hop
        _, rmag = oracle.stft_f32(audios[b], win, fft, hop)
        assert rel_err(mag[b, :T], rmag) < TOL_STFT, b
        assert not mag[b, T:].any() and not H1[b, T:].any(), b             # padding frames are zero and stay zero
        rW, rH, _, _ = oracle.nmf_process(rmag, K, iters, uw, True, 42 if seeds is None else seeds[b],
                                          W0=None if seedW is None else seedW[b].astype(np.float64))
        assert rel_err(W1[b], rW) < TOL_FACTORS_TIGHT and rel_err(H1[b, :T], rH) < TOL_FACTORS_TIGHT, b
        rb, ra = oracle.bufnmf_writeback(rW, rH)
        assert bases[b].shape == rb.shape and acts[b].shape == ra.shape
        assert rel_err(bases[b], rb) < 1e-6 and rel_err(acts[b], ra) < 1e-6, b
    return plan


def test_ragged_corpus_few_buffers_split_contractions(ctx, oracle, onp):
    """fluhip_corpus_create_ragged, the small-corpus regime: a handful of buffers of very different lengths (one frame to 12 s)
    in one set of launches -- the W update's contractions split by each buffer's own length (work-list mode + the finalize's
    per-buffer split table), the H update's strips following each buffer's frames; every buffer against the oracle"""
    lens = [529200, 100, 44100, 200001, 88200, 3000, 352800, 257, 61234]
    plan = _check_ragged(ctx, oracle, onp, lens, 2048, 2048, 512, 32, 8, range(len(lens)))
    assert plan["kernel"] == 5 and plan["split_w"] > 1, plan
    # per-buffer seeds, a rank that is not a multiple of 16, another transform size, seeded and fixed bases
    lens = [30000, 22050, 30000, 5000, 22050, 30000, 12345, 257]
    _check_ragged(ctx, oracle, onp, lens, 1024, 1024, 256, 5, 10, range(len(lens)), seeds=[7, 7, 8, 9, 10, 11, 12, 13])
    rs = np.random.RandomState(2)
    sW = rs.uniform(0.05, 1.0, (len(lens), 5, 513)).astype(np.float32)
    _check_ragged(ctx, oracle, onp, lens, 1024, 1024, 256, 5, 10, (0, 3, 7), seedW=sW)
    _check_ragged(ctx, oracle, onp, lens, 1024, 1024, 256, 5, 10, (1, 6), seedW=sW, uw=False)
    _check_ragged(ctx, oracle, onp, [40000, 9000, 70001], 4096, 4096, 1024, 70, 4, range(3))   # padded rank 128, fft 4096


def test_ragged_corpus_activation_seeds_and_resynthesis(ctx, oracle, onp):
    """the rest of the BufNMF parameter set on a ragged corpus: fixed activations (count x K x T with T the longest
    buffer's frames; entries past a buffer's own are ignored) and the resynthesis output, against the single-channel
    entry point buffer by buffer"""
    import fluhip
    lens = [12000, 5000, 20001, 257]
    win, fft, hop, K, iters = 1024, 1024, 256, 4, 8
    audios = [onp.synth_audio(n, 9300 + i) for i, n in enumerate(lens)]
    c = fluhip.RaggedCorpus(ctx, lens, win, fft, hop, K)
    rs = np.random.RandomState(8)
    sH = rs.uniform(0.05, 1.0, (len(lens), K, c.T)).astype(np.float32)
    c.keep_spectrum(True)
    c.set_audio(audios); c.stft()
    c.set_factors(None, sH)
    c.nmf(iters, seed=42, updateH=False)
    bases, acts = c.writeback()
    res = c.resynth()
    c.close()
    for b, n in enumerate(lens):
        T = (n + hop) // hop
        rb, ra, rr, rc = ctx.bufnmf_channel(audios[b], win, fft, hop, K, iters, 42, updateH=False,
                                            acts_seed=np.ascontiguousarray(sH[b, :, :T]), resynth=True)
        assert rc == 0 and rel_err(bases[b], rb) < 1e-6 and rel_err(acts[b], ra) < 1e-6, b
        assert res[b].shape == (K, n) and rel_err(res[b], rr) < 1e-6, b


def test_ragged_corpus_many_buffers_whole_contractions(ctx, oracle, onp):
    """the large-corpus regime: enough buffers that every (buffer, strip) wavefront keeps its whole contraction -- no
    split, results and column statistics straight from the update kernel, the Nyquist side column -- with lengths from
    0.1 s to 3 s dealt longest first"""
    rs = np.random.RandomState(4)
    lens = [int(x) for x in rs.randint(4410, 132300, 260)]
    lens[17] = 132300; lens[200] = 300
    plan = _check_ragged(ctx, oracle, onp, lens, 2048, 2048, 512, 32, 6, (0, 17, 99, 200, 259))
    assert plan["split_w"] == 1 and plan["side_column"] == 1, plan


def test_ragged_corpus_rejects_what_it_does_not_cover(ctx):
    import fluhip
    for args in (([1000, 2000], 1024, 1024, 256, 200),      # rank above 128
                 ([1000, 2000], 512, 512, 128, 4),          # no block STFT for fft 512
                 ([1000, 0], 1024, 1024, 256, 4)):          # an empty buffer
        with pytest.raises(fluhip.FluhipError):
            fluhip.RaggedCorpus(ctx, *args)


@pytest.mark.parametrize("uw,uh", [(True, True), (False, True), (True, False), (False, False)])
def test_corpus_seeded_and_fixed_factors(ctx, oracle, onp, uw, uh):
    """fluhip_corpus_set_factors: basesMode / actMode Seed and Fixed (nrt/NMFClient.hpp:246-258 -> alg/NMF.hpp:102-124) on
    the batched form -- per-buffer seeds for W and H, every update combination incl. both fixed (0 iterations of work:
    clamp + normalise only, :150-153), every buffer against the oracle; and clearing the seeds goes back to random draws"""
    import fluhip
    B, n, win, fft, hop, K, iters = 5, 16000, 1024, 1024, 256, 6, 12
    audio = np.stack([onp.synth_audio(n, 7000 + b) for b in range(B)])
    c = fluhip.Corpus(ctx, B, n, win, fft, hop, K)
    rs = np.random.RandomState(11)
    sW = rs.uniform(0.05, 1.0, (B, K, c.F)).astype(np.float32)
    sH = rs.uniform(0.05, 1.0, (B, K, c.T)).astype(np.float32)
    c.set_audio(audio); c.stft()
    mag = c.read_f64(factors=False)[0]
    c.set_factors(sW, sH)
    c.nmf(iters, seed=42, updateW=uw, updateH=uh)
    _, W1, H1 = c.read_f64(mag=False)
    bases, acts = c.writeback()
    for b in range(B):
        rW, rH, _, _ = oracle.nmf_process(mag[b], K, iters, uw, uh, 42, W0=sW[b].astype(np.float64),
                                          H0=np.ascontiguousarray(sH[b].T.astype(np.float64)))
        assert rel_err(W1[b], rW) < TOL_FACTORS_TIGHT and rel_err(H1[b], rH) < TOL_FACTORS_TIGHT, b
        rb, ra = oracle.bufnmf_writeback(rW, rH)
        assert rel_err(bases[b], rb) < 1e-6 and rel_err(acts[b], ra) < 1e-6
    # only W seeded, H random from the seed; then no seeds at all
    c.set_factors(sW, None)
    c.nmf(iters, seed=5063, updateW=uw, updateH=uh)
    _, W1, H1 = c.read_f64(mag=False)
    rW, rH, _, _ = oracle.nmf_process(mag[3], K, iters, uw, uh, 5063, W0=sW[3].astype(np.float64))
    assert rel_err(W1[3], rW) < TOL_FACTORS_TIGHT and rel_err(H1[3], rH) < TOL_FACTORS_TIGHT
    c.set_factors(None, None)
    c.nmf(iters, seed=5063, updateW=uw, updateH=uh)
    _, W1, H1 = c.read_f64(mag=False)
    rW, rH, _, _ = oracle.nmf_process(mag[1], K, iters, uw, uh, 5063)
    assert rel_err(W1[1], rW) < TOL_FACTORS_TIGHT and rel_err(H1[1], rH) < TOL_FACTORS_TIGHT
    c.close()


@pytest.mark.parametrize("name,n,win,fft,hop,K,marks", [("c2", 2646000, 2048, 2048, 512, 16, (20, 60, 200)),
                                                       ("c3", 26460000, 4096, 4096, 1024, 128, (12, 120, 500))])
def test_single_buffer_full_size_properties(ctx, onp, name, n, win, fft, hop, K, marks):
    """BASELINE configs 2 and 3 at their full per-channel size AND their full iteration counts (60 s rank 16, 200
    iterations; 10 min rank 128, 500 iterations -- the long-iteration regime: entries decaying to eps, the deferred
    normalisation, split partials), through size-independent properties: frame count, unit-norm dictionary columns,
    non-negativity, finiteness, KL divergence not increasing from mark to mark (c3: 12 / 120 / 500 iterations),
    bit-identical repeat of the whole run (tests/algorithms/public/TestNMF.cpp:31-39)."""
    import fluhip
    base = onp.synth_audio(441000, 1000)
    x = np.tile(base, n // len(base) + 1)[:n].astype(np.float32)
    c = fluhip.Corpus(ctx, 1, n, win, fft, hop, K)
    assert (c.T, c.F) == ((n + hop) // hop, fft // 2 + 1)
    c.set_audio(x[None, :]); c.stft()

    def kl_rows(V, W1, H1, rows):
        P = np.maximum(H1[rows] @ W1, 1e-300)
        Vs = np.maximum(V[rows], 1e-300)
        return float((V[rows] * np.log(Vs / P) - V[rows] + P).sum())

    mag = c.read_f64(mag=True, factors=False)[0]
    rows = np.arange(0, mag.shape[1], max(1, mag.shape[1] // 2000))   # a frame subset keeps the host side light
    kls = []
    for it in marks:
        c.nmf(it, seed=42)
        _, Wb, Hb = c.read_f64(mag=False)
        assert np.isfinite(Wb).all() and np.isfinite(Hb).all() and (Wb >= 0).all() and (Hb >= 0).all(), it
        assert np.allclose(np.sqrt((Wb[0] * Wb[0]).sum(axis=1)), 1.0, atol=1e-12), it
        kls.append(kl_rows(mag[0], Wb[0], Hb[0], rows))
    c.nmf(marks[-1], seed=42)
    _, Wc, Hc = c.read_f64(mag=False)
    c.close()
    assert np.array_equal(Wb, Wc) and np.array_equal(Hb, Hc)
    print(name, "KL over a frame subset at", marks, "iterations:", kls)
    for a, b in zip(kls, kls[1:]):
        assert b <= a * (1 + 1e-9), kls


def test_bench_workload_matches_oracle_at_full_size(ctx, oracle, onp):
    """the bench.py workload itself (BASELINE config 4 shard: 128 x 10 s, fft 2048 / hop 512, rank 32, 200 iterations,
    seed 42 -- the schedule bench.py times: deferred normalisation, side column, 8 strips): two of its buffers
    against the oracle run on the same audio for all 200 iterations"""
    import fluhip
    B, n, win, fft, hop, K, iters = 128, 441000, 2048, 2048, 512, 32, 200
    distinct = [onp.synth_audio(n, 1000 + b) for b in range(2)]
    audio = np.stack([distinct[b % 2] for b in range(B)])
    c = fluhip.Corpus(ctx, B, n, win, fft, hop, K)
    plan = c.plan()
    assert plan["deferred_norm"] == 1 and plan["side_column"] == 1 and plan["strips_w"] == 8, plan
    c.set_audio(audio); c.stft(); c.nmf(iters, seed=42)
    mag, W1, H1 = c.read_f64()
    bases, acts = c.writeback()
    c.close()
    # every replica of the two inputs, bit for bit (a timing-dependent fault shows in a few buffers of a full chip, not in
    # the two that meet the oracle below)
    for b in range(2, B):
        assert np.array_equal(W1[b], W1[b % 2]) and np.array_equal(H1[b], H1[b % 2]), b
        assert np.array_equal(bases[b], bases[b % 2]) and np.array_equal(acts[b], acts[b % 2]), b
    for b in (0, 127):
        _, rmag = oracle.stft_f32(audio[b], win, fft, hop)
        assert rel_err(mag[b], rmag) < TOL_STFT
        rW, rH, _, _ = oracle.nmf_process(rmag, K, iters, True, True, 42)
        assert rel_err(W1[b], rW) < TOL_FACTORS_TIGHT and rel_err(H1[b], rH) < TOL_FACTORS_TIGHT
        # north_star's bar, element by element ("W/H within 1e-5 relative"; tests/helpers.py elementwise_rel_err)
        assert elementwise_rel_err(W1[b], rW) < TOL_FACTORS and elementwise_rel_err(H1[b], rH) < TOL_FACTORS
        rb, ra = oracle.bufnmf_writeback(rW, rH)
        assert rel_err(bases[b], rb) < 1e-6 and rel_err(acts[b], ra) < 1e-6


def test_corpus_c4_shape_properties(ctx, onp):
    """BASELINE config 4 shape at full per-buffer size (10 s, fft 2048, rank 32, 200 iterations),
    checked through size-independent properties: unit-norm dictionary columns, non-negativity,
    activations' max exactly 1 after write-back, monotone KL divergence between 50 and 200
    iterations, bit-identical repeat."""
    import fluhip
    B, n, win, fft, hop, K = 8, 441000, 2048, 2048, 512, 32
    audio = np.stack([onp.synth_audio(n, 1000 + b) for b in range(B)])
    c = fluhip.Corpus(ctx, B, n, win, fft, hop, K)
    assert (c.T, c.F) == (862, 1025)
    c.set_audio(audio); c.stft()

    def kl(V, W1, H1):
        P = np.maximum(H1 @ W1, 1e-300)
        Vs = np.maximum(V, 1e-300)
        return float((V * np.log(Vs / P) - V + P).sum())

    c.nmf(50, seed=42)
    mag, W50, H50 = c.read_f64()
    c.nmf(200, seed=42)
    _, W200, H200 = c.read_f64()
    bases, acts = c.writeback()
    c.nmf(200, seed=42)
    _, W200b, H200b = c.read_f64()
    c.close()
    assert np.array_equal(W200, W200b) and np.array_equal(H200, H200b)
    assert (W200 >= 0).all() and (H200 >= 0).all() and np.isfinite(W200).all() and np.isfinite(H200).all()
    assert np.allclose(np.sqrt((W200 * W200).sum(axis=2)), 1.0, atol=1e-12)
    assert np.allclose(acts.reshape(B, -1).max(axis=1), 1.0, atol=1e-6)
    for b in range(B):
        assert kl(mag[b], W200[b], H200[b]) <= kl(mag[b], W50[b], H50[b]) * (1 + 1e-9)


# ---------------------------------------------------------------------------------------
# resynthesis: NMF::estimate -> RatioMask -> ISTFT, nrt/NMFClient.hpp:302-334
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,win,fft,hop,K", [(20000, 1024, 1024, 256, 3), (9000, 512, 1024, 128, 2),
                                             (44100, 2048, 2048, 512, 4), (5000, 64, 64, 16, 2),
                                             (30001, 2048, 2048, 256, 3), (30001, 2048, 2048, 1024, 9),   # the batched
                                             (1500, 2048, 2048, 512, 2), (20001, 1024, 1024, 128, 4),     # kernel's hops
                                             (20001, 1024, 1024, 512, 10), (15000, 1024, 2048, 256, 3),    # win < fft
                                             (15000, 512, 2048, 512, 4), (7000, 256, 1024, 128, 9)])
def test_resynthesis_vs_oracle(ctx, oracle, onp, n, win, fft, hop, K):
    x = onp.synth_audio(n, 4242)
    iters = 30
    bases, acts, res, rc = ctx.bufnmf_channel(x, win, fft, hop, K, iters, 42, resynth=True)
    assert rc == 0 and res.shape == (K, n)
    spec, mag = oracle.stft_f32(x, win, fft, hop)
    W1, H1, V1, _ = oracle.nmf_process(mag, K, iters, True, True, 42)
    total = np.zeros(n)
    for k in range(K):
        ref = oracle.resynth_component(spec, W1, H1, V1, k, win, fft, hop, n)
        scale = max(np.abs(ref).max(), 1e-12)
        assert np.abs(res[k] - ref).max() / scale < 1e-5      # float output of an f64 pipeline
        total += res[k]
    # soft masks sum to ~1: the components add back up to the input (away from the edges)
    if n > 3 * win and hop <= win // 2:           # (hop = win: the window's zeros lose samples, nothing adds back up)
        assert np.abs(total[win:-win] - x[win:-win]).max() < 0.02


# ---------------------------------------------------------------------------------------
# feature pipeline (BASELINE config 5): BufMelBands / BufMFCC
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,win,fft,hop", [(88200, 1024, 1024, 512),   # config 5 slice: 173 frames
                                           (30000, 1024, 2048, 256), (20000, 2048, 2048, 512),
                                           (12345, 1000, 1024, 300),    # hop does not divide win
                                           (5024, 301, 512, 75),        # odd window: padding 2 (win >> 1) = win - 1
                                           (5000, 256, 256, 64)])
def test_bufmfcc_vs_oracle(ctx, oracle, onp, n, win, fft, hop):
    audio = np.stack([onp.synth_audio(n, 7000 + b) for b in range(3)])
    got = ctx.bufmfcc(audio, win, fft, hop)
    assert got.shape == (3, 13, 1 + (n + 2 * (win // 2)) // hop - win // hop)
    if (n, win, hop) == (88200, 1024, 512):
        assert got.shape[2] == 173
    for b in range(3):
        ref = oracle.bufmfcc_channel(audio[b], win, fft, hop)
        # dB-scaled cepstra of magnitude O(100): absolute agreement far below audible resolution
        assert np.abs(got[b] - ref).max() < 2e-3, np.abs(got[b] - ref).max()
        assert np.abs(got[b] - ref).max() / np.abs(ref).max() < 1e-5


@pytest.mark.parametrize("padding_mode", [0, 2])
@pytest.mark.parametrize("n,win,fft,hop", [(88200, 1024, 1024, 512), (12345, 1000, 1024, 300), (5024, 301, 512, 75)])
def test_bufmfcc_padding_modes(ctx, onp, n, win, fft, hop, padding_mode):
    """The wrapper's "padding" parameter (None / Full; Default is every other test): frame count and positions of
    StreamingControl (cc/FluidNRTClientWrapper.hpp:557-579, 643-656) against the numpy restatement, whose closed form
    tests/test_oracle.py derives from the FluidSource model for all three modes."""
    audio = np.stack([onp.synth_audio(n, 7100 + b) for b in range(2)])
    got = ctx.bufmfcc(audio, win, fft, hop, padding_mode=padding_mode)
    mb = ctx.bufmelbands(audio, win, fft, hop, padding_mode=padding_mode)
    for b in range(2):
        ref = onp.bufmfcc_channel(audio[b], win, fft, hop, padding_mode=padding_mode)
        assert got[b].shape == ref.shape
        assert np.abs(got[b] - ref).max() < 2e-3
        assert np.abs(got[b] - ref).max() / np.abs(ref).max() < 1e-5
        refm = onp.bufmelbands_channel(audio[b], win, fft, hop, padding_mode=padding_mode)
        assert np.abs(mb[b] - refm).max() / np.abs(refm).max() < 1e-5


def test_bufmfcc_against_the_references_own_corpus_rows(ctx):
    """tests/golden/reference_corpus_mfcc.npz: four slices of flucoma-core's demo corpus with the rows of its
    Resources/Data/flucoma_corpus_mfcc.json -- the mean and deviation of BufMFCC's coefficients 1..13 over each slice, as a
    FluCoMa build computed them (tools/make_reference_mfcc_fixture.py; tests/test_oracle.py holds both oracles against all
    299 recomputable slices).  The HIP path against outputs of the reference itself: measured 1.6e-6 .. 3.8e-6 on values up to
    60, the float32 the JSON stores."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_corpus_mfcc.npz"))
    x = (g["pcm16"].astype(np.float64) / 32768.0).astype(np.float32)
    pts = g["points"]
    for j in range(len(pts) - 1):
        seg = np.ascontiguousarray(x[int(pts[j]):int(pts[j + 1])])
        m = ctx.bufmfcc(seg, 1024, 1024, 512, n_bands=40, n_coefs=13, start_coeff=1)[0].astype(np.float64)
        got = np.concatenate([m.mean(axis=1), m.std(axis=1)])
        assert np.abs(got - g["expected"][j]).max() < 2e-5, (j, np.abs(got - g["expected"][j]).max())


def test_bufmfcc_options(ctx, oracle, onp):
    x = onp.synth_audio(40000, 99)
    for (nb, nc, sc) in ((40, 13, 1), (20, 20, 0), (64, 5, 0), (100, 13, 1)):
        got = ctx.bufmfcc(x, 1024, 1024, 512, n_bands=nb, n_coefs=nc, start_coeff=sc, lo=50.0, hi=12000.0)[0]
        ref = oracle.bufmfcc_channel(x, 1024, 1024, 512, n_bands=nb, n_coefs=nc, start_coeff=sc, lo=50.0, hi=12000.0)
        assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-5


@pytest.mark.parametrize("normalize,scale_db", [(True, False), (False, False), (False, True), (True, True)])
def test_bufmelbands_vs_oracle(ctx, oracle, onp, normalize, scale_db):
    audio = np.stack([onp.synth_audio(30000, 8000 + b) for b in range(2)])
    got = ctx.bufmelbands(audio, 1024, 1024, 512, normalize=normalize, scale_db=scale_db)
    for b in range(2):
        ref = oracle.bufmelbands_channel(audio[b], 1024, 1024, 512, normalize=normalize, scale_db=scale_db)
        if scale_db:
            assert np.abs(got[b] - ref).max() < 2e-3
        else:
            assert np.abs(got[b] - ref).max() / np.abs(ref).max() < 1e-5


def test_bufmfcc_batch_matches_single(ctx, onp):
    audio = np.stack([onp.synth_audio(22050, 9000 + b) for b in range(5)])
    batch = ctx.bufmfcc(audio, 1024, 1024, 512)
    for b in (0, 3, 4):
        assert np.array_equal(batch[b], ctx.bufmfcc(audio[b], 1024, 1024, 512)[0])


# ---------------------------------------------------------------------------------------
# BufSTFT: nrt/BufSTFTClient.hpp forward (mag + phase) and inverse
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("n,win,fft,hop", [(20000, 1024, 1024, 512), (9999, 512, 1024, 100), (50000, 2048, 2048, 512)])
def test_bufstft_forward_inverse(ctx, oracle, onp, mode, n, win, fft, hop):
    x = onp.synth_audio(n, 31)
    mag, ph = ctx.bufstft_forward(x, win, fft, hop, mode)
    rmag, rph = oracle.bufstft_forward(x, win, fft, hop, mode)
    assert mag.shape == rmag.shape                                   # hop count: exact
    assert np.abs(mag - rmag).max() / rmag.max() < 1e-6
    big = rmag > 1e-4 * rmag.max()                                   # phase is ill-conditioned on empty bins
    dphi = np.abs(np.angle(np.exp(1j * (ph.astype(np.float64) - rph))))
    assert dphi[big].max() < 1e-5
    y = ctx.bufstft_inverse(rmag, rph, win, fft, hop, mode)
    ry = oracle.bufstft_inverse(rmag, rph, win, fft, hop, mode)
    assert y.shape == ry.shape
    # std::polar runs in single precision on both sides (nrt/BufSTFTClient.hpp:248-250) and the device's cosf / sinf are
    # not libm's to the last bit: the frames agree to float rounding, and the overlap-add divides that by the summed
    # squared window -- nearly zero on the first and last samples when there is no padding.  Weighted by that divisor:
    w2 = onp.hann(win) ** 2
    nrm = np.zeros((mag.shape[1] - 1) * hop + win)
    for t in range(mag.shape[1]):
        nrm[t * hop:t * hop + win] += w2
    pad = onp.bufstft_padding(win, hop, mode)
    nrm = np.maximum(nrm[pad:pad + len(y)], 2.220446049250313e-16)
    assert (np.abs(y - ry) * np.minimum(nrm, 1.0)).max() < 2e-6 * max(1.0, np.abs(rmag).max() / fft)
    m = min(len(y), n)
    assert np.abs(y[win:m - win] - x[win:m - win]).max() < 1e-5      # round trip away from the edges


# ---------------------------------------------------------------------------------------
# window types of algorithm::WindowFuncs (the STFT entry point takes windowType like STFT::STFT)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("wtype", [0, 1, 2, 3])
def test_stft_window_types(ctx, wtype):
    """alg/WindowFuncs.hpp:41-65: Hann, HannD, Hamming, Blackman-Harris (as written there: the three
    cosine terms share one argument)"""
    rs = np.random.RandomState(wtype)
    x = rs.standard_normal(6000)
    win, fft, hop = 512, 512, 128
    i = np.arange(win)
    arg = 2 * np.pi * i / win
    w = [0.5 - 0.5 * np.cos(arg), (np.pi / win) * np.sin(arg), 0.54 - 0.46 * np.cos(arg),
         0.35875 - 0.48829 * np.cos(arg) + 0.14128 * np.cos(arg) + 0.01168 * np.cos(arg)][wtype]
    spec, mag = ctx.stft(x, win, fft, hop, window_type=wtype)
    padded = np.zeros(len(x) + win + hop)
    padded[win // 2: win // 2 + len(x)] = x
    T = (len(x) + hop) // hop
    idx = np.arange(T)[:, None] * hop + i[None, :]
    ref = np.fft.rfft(padded[idx] * w[None, :], axis=1)
    ref[:, 0] = ref[:, 0].real
    ref[:, -1] = ref[:, -1].real
    assert np.abs(spec - ref).max() / np.abs(ref).max() < 1e-12
    assert np.abs(mag - np.abs(ref)).max() / np.abs(ref).max() < 1e-12


def test_stft_gaussian_window_needs_odd_size(ctx):
    import fluhip
    x = np.zeros(4000)
    with pytest.raises(fluhip.FluhipError):
        ctx.stft(x, 512, 512, 128, window_type=4)      # alg/WindowFuncs.hpp:69 assert(size % 2)
    spec, mag = ctx.stft(x + 1.0, 511, 512, 128, window_type=4)
    assert np.isfinite(mag).all()


@pytest.mark.parametrize("n,win,fft,hop", [(20000, 1024, 1024, 256), (9999, 512, 512, 100), (5024, 301, 512, 75)])
@pytest.mark.parametrize("padding_mode", [0, 1, 2])
def test_feature_frame_positions_follow_fluidsource(ctx, onp, n, win, fft, hop, padding_mode):
    """Frame indexing of the HIP feature path against the restated FluidSource chain
    (tests/clients/common/TestFluidSource.cpp:17-59 pins that restatement): a unit impulse at sample p lights up exactly
    the kept frames whose window covers p (the Hann window is zero at its first sample only)."""
    T, starts = onp.streaming_control_frame_starts(n, win, hop, padding_mode)
    for p in (0, 1, win // 2, n // 3, n - 2, n - 1):
        x = np.zeros(n, dtype=np.float32)
        x[p] = 1.0
        mb = ctx.bufmelbands(x, win, fft, hop, n_bands=8, lo=20.0, hi=20000.0, normalize=False, scale_db=False,
                             padding_mode=padding_mode)[0]
        assert mb.shape[1] == T
        lit = mb.sum(axis=0) > 0
        _, start0 = onp.feature_frames(n, win, hop, padding_mode)
        for k in range(T):
            s = starts[k] if starts[k] is not None else start0 + k * hop
            covered = s < p < s + win          # i = p - s in 1 .. win-1: non-zero window sample
            assert bool(lit[k]) == covered, (p, k, s)


@pytest.mark.parametrize("frame", [32, 64, 256, 1024, 8192])
def test_reference_testbufferedprocess_cola_through_bufstft(ctx, frame):
    """tests/clients/common/TestBufferedProcess.cpp:20-70 through the HIP path: the reference's step signal (a frame of
    zeros, then ones), Hann-windowed frames at hop = frameSize / 2, transformed and resynthesised by BufSTFT forward ->
    inverse.  Under COLA the input comes back (there to 1e-12 in doubles; here through float magnitude / phase buffers)."""
    hop = frame // 2
    n = min(128 * frame, 16 * frame + 4096)
    x = np.zeros(n, dtype=np.float32)
    x[frame:] = 1.0
    mag, ph = ctx.bufstft_forward(x, frame, frame, hop, 1)
    assert mag.shape == (frame // 2 + 1, 1 + (n + 2 * (frame // 2) - frame) // hop)
    y = ctx.bufstft_inverse(mag, ph, frame, frame, hop, 1)
    m = min(len(y), n)
    assert np.abs(y[:m - frame] - x[:m - frame]).max() < 5e-6


@pytest.mark.parametrize("n,win,fft,hop,K,mode", [(30000, 1024, 1024, 512, 5, 1), (12345, 1000, 1024, 300, 3, 0),
                                                  (9000, 512, 1024, 128, 20, 2), (5000, 256, 256, 384, 2, 1),
                                                  (20000, 2048, 2048, 512, 33, 1)])
def test_nmfmatch_vs_oracle(ctx, onp, n, win, fft, hop, K, mode):
    """fluhip_nmfmatch_f32 = NMFMatch (clients/rt/NMFMatchClient.hpp:76-118) behind StreamingControl: every frame of every
    channel as one batch of the H update, against the numpy restatement (itself held against a literal model of the client
    on the CPU): two channels, the three padding modes, hop not dividing the window, hop above the window, win < fft,
    ranks on both sides of the 16 / 32 paddings"""
    rs = np.random.RandomState(K)
    audio = np.stack([onp.synth_audio(n, 8100 + c) for c in range(2)])
    bases = (np.abs(rs.standard_normal((K, fft // 2 + 1))) + 0.01).astype(np.float32)
    got = ctx.nmfmatch(audio, bases, win, fft, hop, seed=42, padding_mode=mode)
    T = onp.feature_frames(n, win, hop, mode)[0]
    assert got.shape == (2, K, T)
    for c in range(2):
        ref = onp.nmfmatch_channel(audio[c], bases, win, fft, hop, 42, mode)
        assert rel_err(got[c], ref) < 1e-5, c
    if win // hop == 0:
        assert not got[:, :, 0].any()


@pytest.mark.parametrize("n,win,fft,hop,K,iters", [(30000, 1024, 1024, 512, 4, 10), (12345, 1000, 1024, 300, 3, 5),
                                                   (9000, 512, 1024, 128, 9, 12), (6000, 256, 256, 256, 2, 3),
                                                   (20000, 2048, 2048, 512, 20, 10)])
def test_nmffilter_vs_oracle(ctx, onp, n, win, fft, hop, K, iters):
    """fluhip_nmffilter_f32 = NMFFilter (clients/rt/NMFFilterClient.hpp:69-118) behind Streaming: processFrame per frame,
    ratio mask per component, inverse frames overlap-added at the ring buffers' positions -- against the numpy
    restatement (held against the literal client model on the CPU); and the components add back up to the input"""
    rs = np.random.RandomState(K + n)
    audio = np.stack([onp.synth_audio(n, 8200 + c) for c in range(2)])
    bases = (np.abs(rs.standard_normal((K, fft // 2 + 1))) + 0.01).astype(np.float32)
    got = ctx.nmffilter(audio, bases, win, fft, hop, iters=iters, seed=42)
    assert got.shape == (2, K, n)
    for c in range(2):
        ref = onp.nmffilter_channel(audio[c], bases, win, fft, hop, iters, 42)
        scale = max(np.abs(ref).max(), 1e-12)
        assert np.abs(got[c] - ref).max() / scale < 1e-5, c
        if hop < win and win % hop == 0:
            assert np.abs(got[c].sum(axis=0) - audio[c]).max() < 1e-4
    with pytest.raises(Exception):
        ctx.nmffilter(audio, bases, 256, 256, 300)


def test_allocation_failures_are_classified_by_code(ctx, onp):
    """the host client retries a batched job channel by channel only after an ALLOCATION failure, and learns that
    from `fluhip_last_error_is_out_of_memory` -- set where the failure happens (hipErrorOutOfMemory here: 0.44 TB of magnitudes
    asked of a 288 GB part, refused by the allocator without touching anything), not from the message text -- cleared by
    `fluhip_clear_error` and by the next failure of another kind; the context stays usable."""
    import fluhip
    lib = ctx.lib
    lib.fluhip_clear_error(ctx.h)
    assert lib.fluhip_last_error_is_out_of_memory(ctx.h) == 0 and lib.fluhip_last_error(ctx.h) == b""
    with pytest.raises(fluhip.FluhipError):
        fluhip.Corpus(ctx, 60000, 441000, 2048, 2048, 512, 32)
    assert lib.fluhip_last_error_is_out_of_memory(ctx.h) == 1
    assert b"out of memory" in lib.fluhip_last_error(ctx.h).lower() or b"hip error" in lib.fluhip_last_error(ctx.h).lower()
    with pytest.raises(fluhip.FluhipError):            # an argument error: not an allocation failure
        fluhip.Corpus(ctx, 4, 441000, 2048, 1000, 512, 32)
    assert lib.fluhip_last_error_is_out_of_memory(ctx.h) == 0
    with pytest.raises(fluhip.FluhipError):
        fluhip.Corpus(ctx, 60000, 441000, 2048, 2048, 512, 32)
    assert lib.fluhip_last_error_is_out_of_memory(ctx.h) == 1
    lib.fluhip_clear_error(ctx.h)
    assert lib.fluhip_last_error_is_out_of_memory(ctx.h) == 0
    x = onp.synth_audio(22050, 1000)
    bases, acts, rc = ctx.bufnmf_channel(x, 1024, 1024, 512, 3, 5, 42)      # the context is as good as before
    assert rc == 0 and np.isfinite(bases).all() and np.isfinite(acts).all()


def test_no_device_memory_is_left_behind(ctx, onp):
    """a host keeps ONE context for its lifetime and runs thousands of jobs through it (clients/common/FluidNRTClientWrapper.hpp:831:
    one client per adaptor): corpora of every schedule family created, run and destroyed in a loop -- plain, split, work lists,
    ragged, off-size rank, the any-rank path, the single-call client entry with resynthesis -- must hand all their device memory
    back (hipMemGetInfo before and after, behind one warm-up pass that fills the context's own caches)"""
    import ctypes
    import fluhip
    hip = ctypes.CDLL("libamdhip64.so")

    def free_bytes():
        f, t = ctypes.c_size_t(0), ctypes.c_size_t(0)
        assert hip.hipDeviceSynchronize() == 0
        assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
        return f.value

    n = 44100
    audio = {B: np.stack([onp.synth_audio(n, 9100 + b) for b in range(min(B, 3))] * ((B + 2) // 3))[:B] for B in (1, 2, 16, 130)}

    def one_pass():
        for B, K in ((130, 32), (16, 40), (2, 100), (1, 16), (1, 150)):
            c = fluhip.Corpus(ctx, B, n, 1024, 1024, 256, K)
            c.set_audio(audio[B]); c.stft(); c.nmf(2, seed=42)
            c.read_f64()
            c.close()
        lens = [n, n // 2, n // 3, 5000, 300]
        r = fluhip.RaggedCorpus(ctx, lens, 1024, 1024, 256, 24)
        r.set_audio([audio[16][i % 16][:m] for i, m in enumerate(lens)]); r.stft(); r.nmf(2, seed=42)
        r.read_f64()
        r.close()
        _, _, res, rc = ctx.bufnmf_channel(audio[1][0], 1024, 1024, 256, 5, 3, 42, resynth=True)
        assert rc == 0 and np.isfinite(res).all()

    one_pass()
    before = free_bytes()
    for _ in range(12):
        one_pass()
    after = free_bytes()
    assert before - after < (8 << 20), (before, after, before - after)
    # and whole contexts (a client that is destroyed takes its context along: NMFClient.hpp ~NMFClient)
    def ctx_pass():
        c2 = fluhip.Context(0, ctx.lib)
        _, _, rc = c2.bufnmf_channel(audio[1][0], 1024, 1024, 256, 3, 2, 42)
        assert rc == 0
        c2.close()
    ctx_pass()
    before = free_bytes()
    for _ in range(6):
        ctx_pass()
    after = free_bytes()
    assert before - after < (8 << 20), (before, after, before - after)


@pytest.mark.parametrize("K", [105, 120, 128])
def test_long_factors_at_rank_128_in_every_update_mode(ctx, oracle, K):
    """Round 5's shortcuts of long single buffers on the arrays of rank 128 (the forms that take their column sums from a
    pre-pass): the side column in front of the W update with its denominators as the column sums of H, the pre-reduced norm
    combine leaving the column sums of the new W' in the H update's denominator slots.  Every pairing of the two updates must
    leave the right sums where the next launch reads them: both factors (alg/NMF.hpp:154-181), W alone against a fixed H, H
    alone against a fixed W, and both from seeded factors -- K = 105 (28 of 32 MFMAs), 120 (zero columns), 128."""
    rs = np.random.RandomState(K)
    T, F, iters = 420, 1025, 7
    X = np.abs(rs.standard_normal((T, 6)) @ rs.standard_normal((6, F))) + 0.01 * rs.uniform(0, 1, (T, F))
    W0 = rs.uniform(0.1, 1.0, (K, F))
    H0 = rs.uniform(0.1, 1.0, (T, K))
    for uw, uh, w0, h0 in ((True, True, None, None), (True, False, None, H0), (False, True, W0, None), (True, True, W0, H0)):
        W1, H1, V1, rc = ctx.nmf_process(X, K, iters, uw, uh, 42, w0, h0)
        rW, rH, rV, _ = oracle.nmf_process(X, K, iters, uw, uh, 42, w0, h0)
        assert rc == 0
        assert rel_err(W1, rW) < TOL_FACTORS_TIGHT and rel_err(H1, rH) < TOL_FACTORS_TIGHT and rel_err(V1, rV) < TOL_FACTORS_TIGHT, \
            (K, uw, uh, rel_err(W1, rW), rel_err(H1, rH))


@pytest.mark.parametrize("K", [4, 16, 20, 24, 32, 33, 40, 48, 56, 64])
@pytest.mark.parametrize("frames", [9, 20])
def test_small_corpora_with_narrow_strips_at_every_compute_rank(ctx, oracle, onp, K, frames):
    """Twenty short buffers: the planner deals each buffer's bins into strips of one to four column groups, the forms whose
    first product sums several partial accumulate chains per group right behind the MFMAs.  Round 6 spelled those chains in
    asm (VGPR results) and the compiler neither padded nor ordered the VALU adds behind them: the W update of ranks 33 .. 40
    at two groups per strip read a chain register two cycles after its MFMA (6e-2 wrong from three steps of four frames on;
    two steps were right -- `tools/isa_mfma_valu_hazard.py`).  Every rank class of the two-operand-set
    pipeline, both updates, three steps and five."""
    import fluhip
    B, win, fft, hop, iters = 20, 2048, 2048, 1024, 3
    n = frames * hop - 5
    src = [onp.synth_audio(n, 8100 + b) for b in range(3)]
    audio = np.stack([src[b % 3] for b in range(B)])
    c = fluhip.Corpus(ctx, B, n, win, fft, hop, K)
    c.set_audio(audio); c.stft()
    for uw, uh in ((True, False), (True, True)):
        c.nmf(iters, seed=42, updateW=uw, updateH=uh)
        mag, W1, H1 = c.read_f64()
        for b in (0, 1, B - 1):
            _, rmag = oracle.stft_f32(audio[b], win, fft, hop)
            rW, rH, _, _ = oracle.nmf_process(rmag, K, iters, uw, uh, 42)
            assert rel_err(W1[b], rW) < TOL_FACTORS_TIGHT and rel_err(H1[b], rH) < TOL_FACTORS_TIGHT, (K, frames, uw, uh, b, c.plan())
    c.close()


@pytest.mark.parametrize("scale", [1e-30, 1e38])
@pytest.mark.parametrize("K", [8, 32, 40, 128])
def test_factor_updates_over_the_magnitude_range_of_float_input(ctx, oracle, K, scale):
    """The quotients' reciprocals come from product trees (csrc/recip_tree.h: one v_rcp_f64 per four operands, six in the
    frame-strip kernel): the root of a tree is a product of up to six clamped Q values, which must stay inside the double range
    for anything a float32 input can produce -- magnitudes up to ~3.4e38 x the window sum -- and for very quiet input, where
    every Q sits far below 1 (alg/NMF.hpp:160, 168 divide element by element and have no such limit)."""
    rs = np.random.RandomState(K)
    T, F, iters = 300, 513, 6
    X = (np.abs(rs.standard_normal((T, 5)) @ rs.standard_normal((5, F))) + 0.01 * rs.uniform(0, 1, (T, F))) * scale
    W1, H1, V1, rc = ctx.nmf_process(X, K, iters, True, True, 42)
    rW, rH, rV, _ = oracle.nmf_process(X, K, iters, True, True, 42)
    assert rc == 0 and np.isfinite(W1).all() and np.isfinite(H1).all()
    assert rel_err(W1, rW) < TOL_FACTORS_TIGHT and rel_err(H1, rH) < TOL_FACTORS_TIGHT and rel_err(V1, rV) < TOL_FACTORS_TIGHT, \
        (K, scale, rel_err(W1, rW), rel_err(H1, rH))


@pytest.mark.parametrize("K", [16, 32])
def test_very_loud_float_audio_through_the_corpus_path(ctx, oracle, onp, K):
    """float32 samples near the top of their range (1e30 x a normal signal): magnitudes ~1e33, the frame-strip kernel at
    rank 16 (six quotients per reciprocal) and the batched kernel at rank 32 (four) against the oracle"""
    import fluhip
    n, win, fft, hop, iters = 30000, 2048, 2048, 512, 5
    x = (onp.synth_audio(n, 4242).astype(np.float64) * 1e30).astype(np.float32)
    assert np.isfinite(x).all()
    c = fluhip.Corpus(ctx, 1, n, win, fft, hop, K)
    c.set_audio(x[None, :]); c.stft(); c.nmf(iters, seed=42)
    mag, W1, H1 = c.read_f64(); plan = c.plan(); c.close()
    _, rmag = oracle.stft_f32(x, win, fft, hop)
    rW, rH, _, _ = oracle.nmf_process(rmag, K, iters, True, True, 42)
    assert rel_err(mag[0], rmag) < 1e-12 and np.isfinite(W1).all() and np.isfinite(H1).all()
    assert rel_err(W1[0], rW) < TOL_FACTORS_TIGHT and rel_err(H1[0], rH) < TOL_FACTORS_TIGHT, (K, plan, rel_err(W1[0], rW), rel_err(H1[0], rH))
