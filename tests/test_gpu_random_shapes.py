"""Seeded sweep over shapes and schedules: whatever plan the library picks for a shape (contraction splits,
deferred normalisation, side column, narrow or wide kernel forms, rank padding) the factors must match the oracle.
Small problems, many of them: the edges are where schedules change (rank 16/17, 64/65, one frame, one bin group,
batches that fill the chip or not)."""
import os

import numpy as np
import pytest

from helpers import TOL_FACTORS_TIGHT, rel_err

pytestmark = pytest.mark.gpu


def _cases():
    # FLUHIP_SWEEP="seed,count" runs a different / larger sweep ad hoc (the committed default is 2024,60 (36 until round 6; ragged 4711,24, was 14))
    seed, count = (int(v) for v in os.environ.get("FLUHIP_SWEEP", "2024,60").split(","))
    rs = np.random.RandomState(seed)
    out = []
    ffts = [64, 128, 256, 512, 1024, 2048]
    ranks = [1, 2, 3, 5, 8, 15, 16, 17, 31, 32, 33, 40, 48, 49, 64, 65, 70, 96, 97, 100, 105, 112, 128]   # (105 / 112: 28 of 32 MFMAs, column sums from a pre-pass like 128)
    for i in range(count):
        fft = ffts[rs.randint(len(ffts))]
        win = fft if rs.rand() < 0.7 else fft // 2
        hop = [win // 2, win // 4, win, win // 2 + 3][rs.randint(4)]
        K = ranks[rs.randint(len(ranks))]
        B = [1, 1, 2, 9, 40, 130][rs.randint(6)]
        frames = [1, 2, 7, 33, 120, 400][rs.randint(6)]
        if K > 64 and (B > 9 or frames > 120 or fft > 512):
            B, frames, fft, win = min(B, 2), min(frames, 60), 256, 256  # keep the oracle's share of the run short
            hop = 128
        n = max(1, frames * hop - rs.randint(0, hop))
        iters = int(rs.randint(0, 9))
        uw, uh = [(True, True), (True, True), (True, False), (False, True)][rs.randint(4)]
        out.append((i, B, n, win, fft, hop, K, iters, uw, uh))
    return out


@pytest.mark.parametrize("case", _cases(), ids=lambda c: "c%d_B%d_n%d_w%d_f%d_h%d_K%d_i%d_%d%d" % c)
def test_random_shape(ctx, oracle, onp, case):
    import fluhip
    _, B, n, win, fft, hop, K, iters, uw, uh = case
    distinct = [onp.synth_audio(n, 7000 + b) for b in range(min(B, 3))]
    audio = np.stack([distinct[b % len(distinct)] for b in range(B)])
    c = fluhip.Corpus(ctx, B, n, win, fft, hop, K)
    assert c.T == (n + hop) // hop and c.F == fft // 2 + 1
    c.set_audio(audio); c.stft()
    c.nmf(iters, seed=42, updateW=uw, updateH=uh)
    mag, W1, H1 = c.read_f64()
    plan = c.plan()
    c.close()
    # every replica of an input, wherever it sits in the corpus: the last window of a corpus may run with another
    # contraction split than the full ones (another summation order, a few 1e-15: tools/probes/r05/replica_diff.py), so
    # the replicas are held to 1e-12 of the first copy rather than to its bits -- the first copy is held to the oracle below
    for b in range(len(distinct), B):
        r = b % len(distinct)
        assert rel_err(W1[b], W1[r]) < 1e-12 and rel_err(H1[b], H1[r]) < 1e-12 and np.array_equal(mag[b], mag[r]), (b, plan)
    for b in sorted({0, B - 1, min(B - 1, 2)}):
        _, rmag = oracle.stft_f32(audio[b], win, fft, hop)
        assert rel_err(mag[b], rmag) < 1e-12, plan
        rW, rH, _, _ = oracle.nmf_process(rmag, K, iters, uw, uh, 42)
        assert rel_err(W1[b], rW) < TOL_FACTORS_TIGHT and rel_err(H1[b], rH) < TOL_FACTORS_TIGHT, plan


def _ragged_cases():
    seed, count = (int(v) for v in os.environ.get("FLUHIP_SWEEP_RAGGED", "4711,24").split(","))
    rs = np.random.RandomState(seed)
    out = []
    for i in range(count):
        fft = [1024, 2048, 4096][rs.randint(3)]
        win = fft if rs.rand() < 0.7 else fft // 2
        hop = [win // 2, win // 4, win][rs.randint(3)]
        K = [1, 3, 16, 17, 32, 40, 64, 100][rs.randint(8)]
        B = [2, 3, 7, 20, 90, 300][rs.randint(6)]
        longest = [1, 5, 40, 200][rs.randint(4)]              # frames of the longest buffer
        if K > 32:
            B, longest = min(B, 20), min(longest, 40)
        lens = [int(max(1, rs.randint(1, longest + 1) * hop - rs.randint(0, hop))) for _ in range(B)]
        iters = int(rs.randint(0, 7))
        uw, uh = [(True, True), (True, True), (True, False), (False, True)][rs.randint(4)]
        out.append((i, B, longest, win, fft, hop, K, iters, uw, uh, tuple(lens)))
    return out


@pytest.mark.parametrize("case", _ragged_cases(), ids=lambda c: "r%d_B%d_T%d_w%d_f%d_h%d_K%d_i%d_%d%d" % c[:10])
def test_random_ragged_corpus(ctx, oracle, onp, case):
    """buffers of random different lengths in one set of launches (work lists, intra-workgroup reduction, per-buffer
    finalize tables): whatever the planner picks, a sample of the buffers -- the shortest and the longest among them --
    must match the oracle, and the padding frames must stay zero"""
    import fluhip
    _, B, _, win, fft, hop, K, iters, uw, uh, lens = case
    lens = list(lens)
    src = [onp.synth_audio(max(lens), 8100 + b) for b in range(3)]
    audios = [src[b % 3][:n] for b, n in enumerate(lens)]
    c = fluhip.RaggedCorpus(ctx, lens, win, fft, hop, K)
    c.set_audio(audios); c.stft()
    c.nmf(iters, seed=42, updateW=uw, updateH=uh)
    mag, W1, H1 = c.read_f64()
    plan = c.plan()
    c.close()
    for b in sorted({0, B - 1, int(np.argmin(lens)), int(np.argmax(lens))}):
        T = (lens[b] + hop) // hop
        _, rmag = oracle.stft_f32(audios[b], win, fft, hop)
        assert rel_err(mag[b, :T], rmag) < 1e-12, plan
        assert not mag[b, T:].any() and not H1[b, T:].any(), plan
        rW, rH, _, _ = oracle.nmf_process(rmag, K, iters, uw, uh, 42)
        assert rel_err(W1[b], rW) < TOL_FACTORS_TIGHT and rel_err(H1[b, :T], rH) < TOL_FACTORS_TIGHT, (b, plan)


def _resynth_cases():
    # FLUHIP_SWEEP_RESYNTH="seed,count": resynthesis shapes (the batched register-overlap-add kernel and, for hops it does not
    # cover, the per-buffer kernels)
    seed, count = (int(v) for v in os.environ.get("FLUHIP_SWEEP_RESYNTH", "99,16").split(","))
    rs = np.random.RandomState(seed)
    out = []
    for i in range(count):
        fft = [1024, 2048, 2048, 512][rs.randint(4)]
        win = fft // [1, 1, 2, 4][rs.randint(4)]
        hop = win // [2, 4, 8, 4, 3, 1][rs.randint(6)]
        K = [1, 2, 3, 7, 8, 9, 16, 17, 33][rs.randint(9)]
        n = int(rs.randint(1, 40)) * hop + int(rs.randint(0, hop))
        out.append((i, n, win, fft, hop, K, int(rs.randint(0, 5))))
    return out


@pytest.mark.parametrize("case", _resynth_cases(), ids=lambda c: "r%d_n%d_w%d_f%d_h%d_K%d_i%d" % c)
def test_random_resynthesis(ctx, oracle, onp, case):
    _, n, win, fft, hop, K, iters = case
    x = onp.synth_audio(max(n, 64), 8100 + n % 97)[:n]
    bases, acts, res, rc = ctx.bufnmf_channel(x, win, fft, hop, K, iters, 42, resynth=True)
    assert rc == 0 and res.shape == (K, n) and np.isfinite(res).all()
    spec, mag = oracle.stft_f32(x, win, fft, hop)
    W1, H1, V1, _ = oracle.nmf_process(mag, K, iters, True, True, 42)
    for k in sorted({0, K - 1, K // 2}):
        ref = oracle.resynth_component(spec, W1, H1, V1, k, win, fft, hop, n)
        assert np.abs(res[k] - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1e-12) + 1e-9, (k, np.abs(res[k] - ref).max())


@pytest.mark.parametrize("K,B,frames,fft", [(64, 130, 33, 256), (48, 40, 120, 256), (128, 130, 33, 256), (100, 40, 120, 256),
                                            (128, 130, 5, 512), (64, 130, 7, 512), (128, 64, 61, 1024), (64, 64, 90, 1024)])
def test_wide_rank_batches(ctx, oracle, onp, K, B, frames, fft):
    """ranks 33..128 in the batched regime (whole strips, the in-place pipeline form with / without the column-sum pre-pass):
    many buffers, few frames -- the corner the random sweep above keeps small for the oracle's sake"""
    import fluhip
    hop = fft // 4
    n = frames * hop - 3
    distinct = [onp.synth_audio(n, 7300 + b) for b in range(3)]
    audio = np.stack([distinct[b % 3] for b in range(B)])
    c = fluhip.Corpus(ctx, B, n, fft, fft, hop, K)
    c.set_audio(audio); c.stft(); c.nmf(3, seed=42)
    mag, W1, H1 = c.read_f64()
    plan = c.plan()
    c.close()
    for b in (0, 1, B - 1):
        _, rmag = oracle.stft_f32(audio[b], fft, fft, hop)
        rW, rH, _, _ = oracle.nmf_process(rmag, K, 3, True, True, 42)
        assert rel_err(W1[b], rW) < TOL_FACTORS_TIGHT and rel_err(H1[b], rH) < TOL_FACTORS_TIGHT, (plan, b)



@pytest.mark.parametrize("K,frames,mode", [(128, 20000, "both"), (100, 17001, "both"), (64, 36000, "both"), (128, 40000, "both"),
                                           (128, 18000, "fixed_w"), (128, 18000, "progress"), (32, 74000, "both")])
def test_two_launch_h_update(ctx, oracle, onp, K, frames, mode):
    """two long buffers at a wide rank: the H update's wavefronts need a poorly filled last round, so it goes out as two
    launches -- whole contractions for the frames that fill whole rounds, split ones for the rest (api_corpus.hip plan_tail);
    also with the bases fixed (no deferred normalisation in the H update) and with a progress callback per iteration"""
    import fluhip
    fft, hop, B, iters = 1024, 256, 2, 3
    n = frames * hop - 5
    audio = np.stack([np.resize(onp.synth_audio(min(n, 2_000_000), 7400 + b), n) for b in range(B)])  # (repeated: the generator is slow)
    c = fluhip.Corpus(ctx, B, n, fft, fft, hop, K)
    seen = []
    c.set_audio(audio); c.stft()
    c.nmf(iters, seed=42, updateW=mode != "fixed_w", progress=(lambda i: seen.append(i) or True) if mode == "progress" else None)
    mag, W1, H1 = c.read_f64()
    plan = c.plan()
    c.close()
    if not os.environ.get("FLUHIP_TAIL_SPLIT") and K > 64:   # (up to rank 64 two buffers this long run on the work lists)
        assert plan["tail_h"] > 1 and plan["split_h"] == 1, plan
    if mode == "progress":
        assert seen == list(range(1, iters + 1)), seen
    for b in range(B):
        _, rmag = oracle.stft_f32(audio[b], fft, fft, hop)
        rW, rH, _, _ = oracle.nmf_process(rmag, K, iters, mode != "fixed_w", True, 42)
        assert rel_err(W1[b], rW) < TOL_FACTORS_TIGHT and rel_err(H1[b], rH) < TOL_FACTORS_TIGHT, (plan, b)


@pytest.mark.parametrize("K,B,seconds", [(32, 1, 100), (20, 1, 70), (64, 1, 60), (32, 2, 40)])
def test_long_buffers_on_the_work_lists(ctx, oracle, onp, K, B, seconds):
    """one or two long buffers at ranks up to 64: past ~45 s of frames in all the planner hands them to the work lists (api_corpus.hip
    list_plan_pays, profiles/r03/plan_regimes.txt) -- the contraction of every strip cut into pieces added up inside a workgroup"""
    import fluhip
    fft, hop, iters = 2048, 512, 4
    n = seconds * 44100
    audio = np.stack([np.resize(onp.synth_audio(1_000_000, 7600 + b), n) for b in range(B)])
    c = fluhip.Corpus(ctx, B, n, fft, fft, hop, K)
    if not os.environ.get("FLUHIP_LIST_PLAN"):
        assert ctx.lib.fluhip_debug_plan_kind(B, c.T, c.F, K) == 1
    c.set_audio(audio); c.stft(); c.nmf(iters, seed=42)
    mag, W1, H1 = c.read_f64()
    plan = c.plan()
    c.close()
    for b in range(B):
        _, rmag = oracle.stft_f32(audio[b], fft, fft, hop)
        rW, rH, _, _ = oracle.nmf_process(rmag, K, iters, True, True, 42)
        assert rel_err(W1[b], rW) < TOL_FACTORS_TIGHT and rel_err(H1[b], rH) < TOL_FACTORS_TIGHT, (plan, b)
