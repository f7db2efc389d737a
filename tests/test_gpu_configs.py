"""The BASELINE configs at their full sizes, as the reference would run them (VERDICT r01 "configs exercised only
in part"): c2 against the oracle, c3 as a stereo job through the threaded host client, c5 at a count that takes
the feature path through several chunks, and bench.py's N > 1 control flow as a bare command."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import TOL_FACTORS, TOL_FACTORS_TIGHT, TOL_STFT, elementwise_rel_err, rel_err  # noqa: F401
from test_client import OK, read_buffer, run

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tiled(onp, n, seed):
    base = onp.synth_audio(441000, seed)
    return np.tile(base, n // len(base) + 1)[:n].astype(np.float32)


def test_c2_full_size_vs_oracle(ctx, oracle, onp):
    """BASELINE config 2 (60 s mono, fft 2048 / hop 512, rank 16) through the client-level entry point, ALL 200 of its
    iterations, against the oracle on the same samples (lean mode: the "ones" GEMMs as sums, ~10 s of one core):
    spectrogram <= 1e-12, factors <= 1e-9 (f64), float outputs."""
    import fluhip
    n, win, fft, hop, K, iters = 2646000, 2048, 2048, 512, 16, 200
    x = _tiled(onp, n, 1000)
    c = fluhip.Corpus(ctx, 1, n, win, fft, hop, K)
    assert (c.T, c.F) == (5168, 1025)
    c.set_audio(x[None, :]); c.stft(); c.nmf(iters, seed=42)
    mag, W1, H1 = c.read_f64()
    bases, acts = c.writeback()
    c.close()
    _, rmag = oracle.stft_f32(x, win, fft, hop)
    assert rel_err(mag[0], rmag) < TOL_STFT
    rW, rH, _, _ = oracle.nmf_process(rmag, K, iters, True, True, 42)
    assert rel_err(W1[0], rW) < TOL_FACTORS_TIGHT and rel_err(H1[0], rH) < TOL_FACTORS_TIGHT
    # north_star's "W/H within 1e-5 relative", element by element (entries above 1e-6 of the largest; tests/helpers.py)
    assert elementwise_rel_err(W1[0], rW) < TOL_FACTORS and elementwise_rel_err(H1[0], rH) < TOL_FACTORS
    rb, ra = oracle.bufnmf_writeback(rW, rH)
    assert rel_err(bases[0], rb) < 1e-6 and rel_err(acts[0], ra) < 1e-6
    # and the one-call client form (fluhip_bufnmf_channel_f32) gives the same floats
    b2, a2, rc = ctx.bufnmf_channel(x, win, fft, hop, K, iters, 42)
    assert rc == 0 and np.array_equal(b2, bases[0]) and np.array_equal(a2, acts[0])


def test_c3_stereo_through_the_threaded_client(driver, oracle, onp, tmp_path, ctx):
    """BASELINE config 3 as the reference runs it: a 10 min STEREO buffer through NRTThreadedNMFClient (async), fft 4096 /
    hop 1024, rank 128 -- a few of its 500 iterations.  Channel 1 (the second pass of the channel loop,
    nrt/NMFClient.hpp:233) is checked against the oracle at full length; both channels by properties."""
    frames, chans = 26460000, 2
    win, hop, fft, K, iters, seed = 4096, 1024, 4096, 128, 2, 42
    audio = np.stack([_tiled(onp, frames, 1000 + c) for c in range(chans)], axis=1)   # frames x chans, interleaved
    inp = tmp_path / "c3.f32"
    audio.tofile(inp)
    prefix = str(tmp_path / "c3")
    r = run(driver, "run", inp, frames, chans, win, hop, fft, K, iters, seed, 0, 0, 1, 0, -1, 0, -1, prefix)
    os.remove(inp)
    assert r["result"] == (OK, "") and r["process"][0] == OK
    bases, sr_b = read_buffer(prefix + "_bases.bin")
    acts, sr_a = read_buffer(prefix + "_acts.bin")
    F, T = fft // 2 + 1, frames // hop + 1
    assert (F, T) == (2049, 25840)
    assert bases.shape == (K * chans, F) and acts.shape == (K * chans, T)
    assert sr_b == pytest.approx(44100.0 / fft) and sr_a == pytest.approx(44100.0 / hop)
    assert np.isfinite(bases).all() and np.isfinite(acts).all() and (bases >= 0).all() and (acts >= 0).all()
    for c in range(chans):
        assert acts[c * K:(c + 1) * K].max() == pytest.approx(1.0, abs=1e-6)           # H / max(H), :289-298
        nrm = np.sqrt((bases[c * K:(c + 1) * K].astype(np.float64) ** 2).sum(axis=1))  # unit dictionary columns, NMF.hpp:162
        assert np.allclose(nrm, 1.0, atol=1e-5)
    x1 = np.ascontiguousarray(audio[:, 1])
    del audio
    rb, ra = oracle.bufnmf_channel(x1, win, fft, hop, K, iters, seed)
    assert rel_err(bases[K:2 * K], rb) < 1e-6 and rel_err(acts[K:2 * K], ra) < 1e-6


def test_c3_decimated_twin_all_iterations_shape(driver, oracle, onp, tmp_path, ctx):
    """the same stereo job on a 12 s twin (fft 4096 / hop 1024, rank 128), ALL 500 iterations of config 3, both channels
    vs the oracle (~30 s of one core per channel): the long-iteration regime at rank 128 -- entries decaying to eps,
    the deferred normalisation carried over 500 updates -- against the restatement, not only by properties"""
    frames, chans = 529200, 2
    win, hop, fft, K, iters, seed = 4096, 1024, 4096, 128, 500, 42
    audio = np.stack([onp.synth_audio(frames, 1000 + c) for c in range(chans)], axis=1)
    inp = tmp_path / "c3s.f32"
    audio.astype(np.float32).tofile(inp)
    prefix = str(tmp_path / "c3s")
    r = run(driver, "run", inp, frames, chans, win, hop, fft, K, iters, seed, 0, 0, 0, 0, -1, 0, -1, prefix)
    assert r["result"] == (OK, "")
    bases, _ = read_buffer(prefix + "_bases.bin")
    acts, _ = read_buffer(prefix + "_acts.bin")
    for c in range(chans):
        rb, ra = oracle.bufnmf_channel(np.ascontiguousarray(audio[:, c]), win, fft, hop, K, iters, seed)
        assert rel_err(bases[c * K:(c + 1) * K], rb) < 1e-6 and rel_err(acts[c * K:(c + 1) * K], ra) < 1e-6
        # element by element on the float outputs, 500 iterations on (north_star: 1e-5 relative)
        assert elementwise_rel_err(bases[c * K:(c + 1) * K], rb) < TOL_FACTORS and elementwise_rel_err(acts[c * K:(c + 1) * K], ra) < TOL_FACTORS


@pytest.mark.parametrize("fused", [1, 0])
def test_c5_multi_chunk_feature_path(ab_ctx, oracle, onp, fused):
    """BASELINE config 5's shape at a count that takes fluhip_bufmfcc_f32 through several chunks of its staging
    buffers (the chunk loop, its per-chunk offsets and synchronisation), in both forms -- the fused STFT -> mel -> DCT
    kernel and the two-kernel form with the magnitudes in HBM: first / boundary / last slices against the oracle,
    every slice against the single-slice call of the same audio.  (FLUHIP_FEAT_CHUNK_BYTES shrinks the chunks: at their
    production size of 2 GiB a test would need tens of thousands of slices to cross one.)"""
    ctx = ab_ctx                                 # the build in which the two switches below are live
    n, win, fft, hop = 88200, 1024, 1024, 512
    distinct = np.stack([onp.synth_audio(n, 1000 + b) for b in range(16)])
    count = 3000
    order = (np.arange(count) * 7) % 16          # neighbours differ, so a wrong chunk offset cannot hide
    audio = distinct[order]
    old = {k: os.environ.get(k) for k in ("FLUHIP_FEAT_CHUNK_BYTES", "FLUHIP_FEAT_FUSED")}
    try:
        os.environ["FLUHIP_FEAT_FUSED"] = str(fused)
        singles = [ctx.bufmfcc(distinct[i][None, :], win, fft, hop)[0] for i in range(16)]
        os.environ["FLUHIP_FEAT_CHUNK_BYTES"] = str(64 << 20)
        out = ctx.bufmfcc(audio, win, fft, hop)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    T = ctx.feature_frames(n, win, hop)
    assert out.shape == (count, 13, T) and T == 173
    for b in range(count):
        assert np.array_equal(out[b], singles[order[b]]), b
    chunk = (64 << 20) // (n * 4) if fused else (64 << 20) // (192 * 544 * 8)
    assert 1 < chunk < count // 3, "the test must cross several chunk boundaries"
    for b in {0, 1, chunk - 1, chunk, chunk + 1, 2 * chunk, count // 2, count - 1}:
        ref = oracle.bufmfcc_channel(audio[b], win, fft, hop)
        assert rel_err(out[b], ref) < 1e-5


def test_c5_production_library_crosses_its_2_gib_chunk(ctx, oracle, onp):
    """the PRODUCTION library's own chunk loop (api_features.hip: staging chunks of 2 GiB, which the test above only reaches in
    the A/B build with shrunken chunks): 6 200 slices of 2 s = 2.19 GB of audio in one fluhip_bufmfcc_f32 call, neighbouring
    slices different, every slice against the single-slice call of its audio, boundary slices against the oracle"""
    n, win, fft, hop = 88200, 1024, 1024, 512
    distinct = np.stack([onp.synth_audio(n, 1000 + b) for b in range(16)])
    count = 6200
    per_chunk = (2 << 30) // (n * 4)
    assert per_chunk < count, "the call must cross the chunk boundary"
    order = (np.arange(count) * 7) % 16
    singles = [ctx.bufmfcc(distinct[i][None, :], win, fft, hop)[0] for i in range(16)]
    out = ctx.bufmfcc(distinct[order], win, fft, hop)
    assert out.shape == (count, 13, 173)
    for b in range(count):
        assert np.array_equal(out[b], singles[order[b]]), b
    for b in (0, per_chunk - 1, per_chunk, per_chunk + 1, count - 1):
        ref = oracle.bufmfcc_channel(distinct[order[b]], win, fft, hop)
        assert rel_err(out[b], ref) < 1e-5, b


def test_c5_device_resident_corpus(ctx, onp):
    """audio and features already in HBM (what a device-side pipeline hands over): same floats as the host-buffer call"""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")          # the runtime the library under test already runs on
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    n, win, fft, hop, count = 88200, 1024, 1024, 512, 96
    audio = np.stack([onp.synth_audio(n, 2000 + b) for b in range(count)])
    ref = ctx.bufmfcc(audio, win, fft, hop)
    a_dev, o_dev = ctypes.c_void_p(), ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(a_dev), audio.nbytes) == 0 and hip.hipMalloc(ctypes.byref(o_dev), ref.nbytes) == 0
    try:
        assert hip.hipMemcpy(a_dev, audio.ctypes.data, audio.nbytes, 1) == 0      # hipMemcpyHostToDevice
        T = ctypes.c_int64(0)
        fp = ctypes.POINTER(ctypes.c_float)
        rc = ctx.lib.fluhip_bufmfcc_f32(ctx.h, ctypes.cast(a_dev, fp), count, n, win, fft, hop, 40, 13, 0, 20.0, 20000.0,
                                        44100.0, ctypes.cast(o_dev, fp), ctypes.byref(T))
        got = np.empty_like(ref)
        assert hip.hipMemcpy(got.ctypes.data, o_dev, ref.nbytes, 2) == 0          # hipMemcpyDeviceToHost
    finally:
        hip.hipFree(a_dev)
        hip.hipFree(o_dev)
    assert rc == 0 and T.value == ref.shape[2]
    assert np.array_equal(got, ref)


def test_pool_two_contexts_shard_a_corpus(ctx, oracle, onp):
    """fluhip_pool_*: the multi-device form behind the C ABI -- here two contexts on device 0, five buffers dealt 3 + 2 by
    one host thread each, results written straight into the caller's arrays in corpus order; every buffer against the
    oracle, progress in order, cancellation"""
    import fluhip
    n, win, fft, hop, K, iters = 30000, 1024, 1024, 256, 5, 12
    audio = np.stack([onp.synth_audio(n, 3000 + b) for b in range(5)])
    pool = fluhip.Pool([0, 0], ctx.lib)
    assert pool.size() == 2 and pool.devices() == [0, 0]
    seen = []
    bases, acts, rc = pool.bufnmf(audio, win, fft, hop, K, iters, seed=42, progress=lambda it: seen.append(it) or True)
    assert rc == 0 and seen == list(range(1, iters + 1))
    for b in range(5):
        rb, ra = oracle.bufnmf_channel(audio[b], win, fft, hop, K, iters, 42)
        assert rel_err(bases[b], rb) < 1e-6 and rel_err(acts[b], ra) < 1e-6
    one = fluhip.Pool([0], ctx.lib)
    b1, a1, _ = one.bufnmf(audio, win, fft, hop, K, iters, seed=42)
    assert rel_err(b1, bases) < 1e-6 and rel_err(a1, acts) < 1e-6
    _, _, rc = pool.bufnmf(audio, win, fft, hop, K, 400, seed=42, progress=lambda it: it < 3)
    assert rc == fluhip.CANCELLED
    pool.close(); one.close()


def test_pool_feature_pipeline_over_two_contexts(ctx, onp):
    """fluhip_pool_bufmfcc_f32 / fluhip_pool_bufmelbands_f32 (BASELINE config 5's multi-device form): seven slices dealt
    4 + 3 over two contexts on device 0; bit-identical to the single-context call (the slices are independent analyses),
    one slice against the numpy restatement, all three padding modes"""
    import fluhip
    n, win, fft, hop = 20000, 1024, 1024, 512
    audio = np.stack([onp.synth_audio(n, 5100 + b) for b in range(7)])
    pool = fluhip.Pool([0, 0], ctx.lib)
    for mode in (1, 0, 2):
        got = pool.bufmfcc(audio, win, fft, hop, padding_mode=mode)
        assert np.array_equal(got, ctx.bufmfcc(audio, win, fft, hop, padding_mode=mode))
        ref = onp.bufmfcc_channel(audio[5], win, fft, hop, padding_mode=mode)
        assert got[5].shape == ref.shape and np.abs(got[5] - ref).max() < 2e-3
    mb = pool.bufmelbands(audio, win, fft, hop, n_bands=24, normalize=False, scale_db=True)
    assert np.array_equal(mb, ctx.bufmelbands(audio, win, fft, hop, n_bands=24, normalize=False, scale_db=True))
    with pytest.raises(fluhip.FluhipError):
        pool.bufmfcc(audio[:, :100], 1024, 1024, 512, n_bands=4000)          # refused by every device, message passed on
    pool.close()


def test_pool_job_with_seeds_fixed_bases_and_resynthesis(ctx, oracle, onp):
    """fluhip_pool_bufnmf_job_f32: the batched form with everything the BufNMF parameter set has -- Seed / Fixed factors
    (basesMode / actMode) and the resynthesis output -- over two contexts on device 0 (shares 3 + 2): fixed bases with
    seeded activations against the oracle per buffer, resynthesis against the single-channel entry point"""
    import fluhip
    n, win, fft, hop, K, iters = 20000, 1024, 1024, 256, 4, 10
    B = 5
    audio = np.stack([onp.synth_audio(n, 3300 + b) for b in range(B)])
    F, T = fft // 2 + 1, (n + hop) // hop
    rs = np.random.RandomState(5)
    sW = rs.uniform(0.05, 1.0, (B, K, F)).astype(np.float32)
    sH = rs.uniform(0.05, 1.0, (B, K, T)).astype(np.float32)
    pool = fluhip.Pool([0, 0], ctx.lib)
    bases, acts, res, rc = pool.bufnmf_job(audio, win, fft, hop, K, iters, seed=42, updateW=False, bases_seed=sW, acts_seed=sH,
                                           resynth=True)
    assert rc == 0
    for b in range(B):
        _, mag = oracle.stft_f32(audio[b], win, fft, hop)
        rW, rH, _, _ = oracle.nmf_process(mag, K, iters, False, True, 42, W0=sW[b].astype(np.float64),
                                          H0=np.ascontiguousarray(sH[b].T.astype(np.float64)))
        rb, ra = oracle.bufnmf_writeback(rW, rH)
        assert rel_err(bases[b], rb) < 1e-6 and rel_err(acts[b], ra) < 1e-6, b
        _, _, r1, _ = ctx.bufnmf_channel(audio[b], win, fft, hop, K, iters, 42, updateW=False, bases_seed=sW[b],
                                         acts_seed=sH[b], resynth=True)
        assert rel_err(res[b], r1) < 1e-6
    pool.close()


def test_pool_large_share_runs_in_pipelined_slices(ctx, oracle, onp):
    """a share of 512 buffers or more without a progress callback goes in slices of 256 whose uploads (copy stream) run
    beside the previous slice's iterations: same floats as the one-corpus path, buffers at the slice edges vs the oracle"""
    import fluhip
    n, win, fft, hop, K, iters = 6000, 1024, 1024, 256, 4, 6
    B = 600
    distinct = [onp.synth_audio(n, 3500 + b) for b in range(7)]
    audio = np.stack([distinct[b % 7] for b in range(B)])
    pool = fluhip.Pool([0], ctx.lib)
    bases, acts, rc = pool.bufnmf(audio, win, fft, hop, K, iters, seed=42)
    assert rc == 0
    for b in (0, 255, 256, 511, 512, 599):
        rb, ra = oracle.bufnmf_channel(audio[b], win, fft, hop, K, iters, 42)
        assert rel_err(bases[b], rb) < 1e-6 and rel_err(acts[b], ra) < 1e-6, b
    seen = []
    b2, a2, rc = pool.bufnmf(audio, win, fft, hop, K, iters, seed=42, progress=lambda it: seen.append(it) or True)   # one corpus
    assert rc == 0 and seen == list(range(1, iters + 1))
    assert rel_err(b2, bases) < 1e-6 and rel_err(a2, acts) < 1e-6
    pool.close()


def test_pool_ragged_corpus(ctx, oracle, onp):
    """fluhip_pool_bufnmf_ragged_f32: buffers of different lengths (a folder of sound files) over two contexts on device 0 --
    dealt by frame count, runs of equal length as one corpus (the batched kernels), the rest one by one (the single-buffer
    schedules); every buffer against the oracle, per-buffer seeds, progress = buffers finished, cancellation"""
    import fluhip
    win, fft, hop, K, iters = 1024, 1024, 256, 5, 10
    lens = [30000, 22050, 30000, 5000, 22050, 30000, 12345, 257]
    audios = [onp.synth_audio(n, 4000 + i) for i, n in enumerate(lens)]
    pool = fluhip.Pool([0, 0], ctx.lib)
    seen = []
    bases, acts, rc = pool.bufnmf_ragged(audios, win, fft, hop, K, iters, seed=42, progress=lambda d: seen.append(d) or True)
    assert rc == 0 and seen == list(range(1, len(lens) + 1)), seen
    for i, a in enumerate(audios):
        rb, ra = oracle.bufnmf_channel(a, win, fft, hop, K, iters, 42)
        assert bases[i].shape == rb.shape and acts[i].shape == ra.shape
        assert rel_err(bases[i], rb) < 1e-6 and rel_err(acts[i], ra) < 1e-6, i
    seeds = [7, 7, 8, 9, 10, 11, 12, 13]
    bases, acts, rc = pool.bufnmf_ragged(audios, win, fft, hop, K, iters, seeds=seeds)
    for i in (0, 2, 7):
        rb, ra = oracle.bufnmf_channel(audios[i], win, fft, hop, K, iters, seeds[i])
        assert rel_err(bases[i], rb) < 1e-6 and rel_err(acts[i], ra) < 1e-6, i
    _, _, rc = pool.bufnmf_ragged(audios, win, fft, hop, K, iters, progress=lambda d: d < 2)
    assert rc == fluhip.CANCELLED
    with pytest.raises(fluhip.FluhipError):
        pool.bufnmf_ragged([np.zeros(0, dtype=np.float32)], win, fft, hop, K, iters)
    pool.close()


def test_ragged_corpus_64_buffers_of_40_lengths_against_the_oracle(ctx, oracle, onp):
    """a folder of different-length files on the batched schedule.  64 buffers of 40 distinct lengths
    (4 .. 16 s, rank 32, fft 2048) as ONE ragged corpus: a sample of the buffers against the oracle (spectrogram, W, H) and
    the plan the work-list scheduler chose.  The SPEED of this shape against its equal-length twin is not a test (round 5: a
    single host-timed sample turned the driver's suite red): it is a row of `tools/perf_matrix.py`, device-timed, min of
    five, recorded per round under `profiles/rNN/perf_matrix.json`."""
    import fluhip
    win, fft, hop, K = 2048, 2048, 512, 32
    rs = np.random.RandomState(7)
    distinct = sorted(int(x) for x in rs.randint(4 * 44100, 16 * 44100, 40))
    lens = [distinct[i % 40] for i in range(64)]
    rs.shuffle(lens)
    assert len(set(lens)) == 40
    base = [onp.synth_audio(16 * 44100, 5000 + i) for i in range(8)]
    audios = [base[i % 8][:n] for i, n in enumerate(lens)]
    c = fluhip.RaggedCorpus(ctx, lens, win, fft, hop, K)
    c.set_audio(audios); c.stft()
    iters = 6
    c.nmf(iters, seed=42)
    mag, W1, H1 = c.read_f64()
    for b in (0, 13, 40, 63, int(np.argmin(lens)), int(np.argmax(lens))):
        T = c.Ts[b]
        _, rmag = oracle.stft_f32(audios[b], win, fft, hop)
        assert rel_err(mag[b, :T], rmag) < TOL_STFT
        rW, rH, _, _ = oracle.nmf_process(rmag, K, iters, True, True, 42)
        assert rel_err(W1[b], rW) < TOL_FACTORS_TIGHT and rel_err(H1[b, :T], rH) < TOL_FACTORS_TIGHT, b
    print(f"64 buffers, 40 lengths: plan {c.plan()}")
    c.close()


def _bench(args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                       timeout=900, env=e, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_two_ranks_as_a_bare_command():
    """`python bench.py --gpus 2` with no launcher around it: bench.py re-executes itself through
    torch.distributed.run, the two ranks shard the corpus (2 + 2 buffers), gather the results, rank 0 prints one line
    with n_gpus = 2 -- and the gathered dictionaries / activations are those of one rank running all 4 buffers.  On a
    box with fewer GPUs than ranks the ranks share the device and the gather goes through gloo; with >= 2 GPUs this is
    the RCCL path itself."""
    common = ["--steps", "2", "--warmup", "1", "--iters", "3", "--no-cpu-baseline"]
    two = _bench(["--gpus", "2", "--buffers", "2", *common])
    one = _bench(["--gpus", "1", "--buffers", "4", *common])
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1 and two["scaling"] == "weak"
    assert two["total_buffers"] == one["total_buffers"] == 4
    assert two["result_finite"] and one["result_finite"]
    # order-sensitive checksum of the gathered floats; the two runs schedule their kernels differently (2 vs 4
    # buffers per launch), which moves f64 sums by rounding only
    assert two["result_checksum"] == pytest.approx(one["result_checksum"], rel=1e-6)
    assert two["value"] > 0 and two["steps"] == 2


def test_bench_eight_ranks_rehearsed_on_one_gpu():
    """The 8-rank job of BASELINE config 4 rehearsed on the one GPU that exists: `python bench.py --gpus 8`
    as the driver's SCALE run issues it, ranks sharing device 0 over gloo -- rendezvous on 127.0.0.1, eight processes each
    synthesising and uploading its own shard, the gather into buffers allocated once outside the steps -- against one rank
    running the same 32 buffers: same order-sensitive checksum, rank r holds shard_range(32, 8, r)."""
    sys.path.insert(0, os.path.join(ROOT, "flucoma-core_amd"))
    import sharding
    common = ["--steps", "1", "--warmup", "1", "--iters", "3", "--no-cpu-baseline"]
    eight = _bench(["--gpus", "8", "--buffers", "4", *common], env={"FLUHIP_BENCH_BACKEND": "gloo"})
    one = _bench(["--gpus", "1", "--buffers", "32", *common])
    assert eight["n_gpus"] == 8 and eight["total_buffers"] == one["total_buffers"] == 8 * 4
    assert eight["shard_ranges"] == [list(sharding.shard_range(32, 8, r)) for r in range(8)]
    assert one["shard_ranges"] == [[0, 32]]
    assert eight["result_finite"] and "gloo" in eight["backend"]
    assert eight["result_checksum"] == pytest.approx(one["result_checksum"], rel=1e-6)
    assert "configs" not in eight and "configs" not in one        # the other configs ride on the default headline only


def test_bench_one_rank_rccl_group_executes_the_collectives():
    """RCCL on the box that is there: FLUHIP_BENCH_BACKEND=nccl at N = 1 builds a ONE-rank `nccl`
    process group and runs the N > 1 job's collective code as written -- all_gather_into_tensor of the device-resident
    dictionaries / activations inside every step, the MAX all-reduce of the step time on a device tensor, the barriers --
    with no `world > 1` short cut.  Same floats as the plain single-process run."""
    common = ["--gpus", "1", "--buffers", "4", "--steps", "2", "--warmup", "1", "--iters", "3", "--no-cpu-baseline"]
    grp = _bench(common, env={"FLUHIP_BENCH_BACKEND": "nccl"})
    one = _bench(common)
    assert grp["backend"].startswith("rccl") and "one-rank group" in grp["backend"] and one["backend"] == "single process"
    assert grp["n_gpus"] == 1 and grp["result_finite"]
    assert grp["result_checksum"] == one["result_checksum"]          # same kernels, same schedule: bit-equal floats
    # the box-invariant fields of the roofline (shader cycles per launch, sustained clock) are there and sane
    r = grp["roofline"]
    assert r["shader_cycles_per_launch"] > 0 and 500 < r["sustained_mhz"] < 2600
    assert r["clock_stamps"]["w"]["launches"] == r["clock_stamps"]["h"]["launches"] == 3 * 3   # 2 timed + 1 profiled step


def _rccl_worker(q, root):
    import socket
    sys.path.insert(0, os.path.join(root, "flucoma-core_amd"))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    import torch
    import torch.distributed as dist
    import sharding
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    loc_b = torch.arange(3 * 5 * 7, dtype=torch.float32, device="cuda").reshape(3, 5, 7)
    gb = sharding.gather_results(loc_b, dist, 1)            # all_gather_into_tensor over RCCL, device tensors
    t = torch.tensor([1.25], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    torch.cuda.synchronize()
    q.put((gb.data_ptr() != loc_b.data_ptr(), bool(torch.equal(gb, loc_b)), float(t.item()), dist.get_backend()))
    dist.destroy_process_group()


def test_rccl_gather_on_one_gpu():
    """sharding.gather_results + the MAX all-reduce exactly as bench.py issues them, on a 1-rank nccl (= RCCL) group: librccl
    loads, a communicator is built on the box's GPU and both collectives run on device tensors."""
    import torch.multiprocessing as mp
    mctx = mp.get_context("spawn")
    q = mctx.Queue()
    p = mctx.Process(target=_rccl_worker, args=(q, ROOT))
    p.start()
    fresh, same, t, backend = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert fresh and same and t == 1.25 and backend == "nccl"


def test_update_clock_stamps(ctx, onp):
    """fluhip_corpus_update_clocks: one wavefront per update launch adds its shader cycles and 100 MHz ticks; the counts
    follow the launches, the implied clock is a plausible shader clock, reset clears"""
    import fluhip
    n, win, fft, hop, K, iters = 44100, 2048, 2048, 512, 32, 6
    audio = np.stack([onp.synth_audio(n, 6000 + b) for b in range(16)])
    c = fluhip.Corpus(ctx, 16, n, win, fft, hop, K)
    c.set_audio(audio); c.stft()
    c.update_clocks(reset=True)
    c.nmf(iters, seed=42)
    k = c.update_clocks(reset=True)
    assert k["w"]["launches"] == iters and k["h"]["launches"] == iters
    for side in ("w", "h"):
        assert k[side]["cycles_per_launch"] > 1000 and 500 < k[side]["sustained_mhz"] < 2600
    z = c.update_clocks()
    assert z["w"]["launches"] == 0 and z["h"]["shader_cycles"] == 0
    c.close()


def test_ctx_trim_returns_cached_blocks(ctx, onp):
    """fluhip_ctx_trim: the per-device cache of freed device blocks is handed back to the driver on request (for hosts that
    share the GPU with other allocators); the next call simply allocates again and gives the same result"""
    x = onp.synth_audio(44100, 5)
    a = ctx.bufnmf_channel(x, 1024, 1024, 256, 3, 5, 42)
    assert ctx.lib.fluhip_ctx_trim(ctx.h) == 0
    b = ctx.bufnmf_channel(x, 1024, 1024, 256, 3, 5, 42)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
