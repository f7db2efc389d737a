"""The frame-strip schedule of a single large buffer at rank <= 16 (kernels_nmf_strip.hip): forced on with FLUHIP_STRIP=1
(read once per process, so in a subprocess) over shapes with one, a few and more than six frame quads per workgroup,
the update-flag combinations of alg/NMF.hpp:154-181, seeded factors, a batched corpus, progress / cancellation, and
same-seed bit identity (tests/algorithms/public/TestNMF.cpp:31-39).  The schedule picked on its own for BASELINE
config 2 is covered at full size in test_gpu_configs.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import os, sys
import numpy as np
ROOT = sys.argv[1]
sys.path.insert(0, os.path.join(ROOT, "flucoma-core_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fluhip, oracle_c, oracle_np
from helpers import rel_err
o = oracle_c.get("native")
ctx = fluhip.Context(0)
worst = 0.0
WANT = 3 if os.environ.get("FLUHIP_STRIP_TILE") == "1" else (2 if os.environ.get("FLUHIP_STRIP_BIN") == "1" else 1)
if os.environ.get("STRIP_TEST_PRODUCTION") == "1":
    WANT = 0   # the production library takes the schedule for SINGLE buffers on its own; batches stay with the batched kernels
def data(T, F):
    rs = np.random.RandomState(T * 7 + F)
    return np.abs(rs.standard_normal((T, 3)) @ rs.standard_normal((3, F))) + 0.01 * rs.uniform(0, 1, (T, F))
for (T, F, K, iters) in ((200, 1025, 16, 12), (33, 17, 1, 10), (300, 513, 3, 10), (1000, 257, 16, 7), (2601, 513, 9, 5),
                         (7003, 129, 16, 4), (5, 33, 2, 3), (1031, 1025, 13, 6),
                         # 35 bin pairs is the most whose W image fits the LDS (F <= 1120); 36 pairs (ADVICE r02: the launch
                         # failed there) must fall back to the split schedule also with the strip schedule forced
                         (150, 1120, 16, 5), (150, 1121, 16, 5), (150, 1152, 9, 5)):
    X = data(T, F)
    W1, H1, V1, rc = ctx.nmf_process(X, K, iters, True, True, 42)
    rW, rH, rV, _ = o.nmf_process(X, K, iters, True, True, 42)
    e = max(rel_err(W1, rW), rel_err(H1, rH), rel_err(V1, rV))
    print("shape", T, F, K, iters, "err", e)
    worst = max(worst, e)
# update-flag combinations and seeded factors
X = data(700, 513)
rs = np.random.RandomState(5)
W0 = rs.uniform(0.01, 1, (7, 513)); H0 = rs.uniform(0.01, 1, (700, 7))
for (uw, uh, w0, h0) in ((True, False, None, None), (False, True, None, None), (False, False, None, None), (True, True, W0, None),
                         (True, True, None, H0), (False, True, W0, H0), (True, False, W0, H0)):
    W1, H1, V1, rc = ctx.nmf_process(X, 7, 6, uw, uh, 11, W0=w0, H0=h0)
    rW, rH, rV, _ = o.nmf_process(X, 7, 6, uw, uh, 11, W0=w0, H0=h0)
    e = max(rel_err(W1, rW), rel_err(H1, rH), rel_err(V1, rV))
    print("flags", uw, uh, w0 is not None, h0 is not None, "err", e)
    worst = max(worst, e)
# zero iterations
W1, H1, V1, rc = ctx.nmf_process(X, 7, 0, True, True, 3)
rW, rH, rV, _ = o.nmf_process(X, 7, 0, True, True, 3)
worst = max(worst, rel_err(W1, rW), rel_err(H1, rH))
# same seed twice: bit-identical
A = ctx.nmf_process(data(2601, 513), 9, 5, True, True, 42)
B = ctx.nmf_process(data(2601, 513), 9, 5, True, True, 42)
assert np.array_equal(A[0], B[0]) and np.array_equal(A[1], B[1])
# progress: every iteration once, in order; a refusal stops the job with the factors of an iteration it reached
seen = []
W1, H1, V1, rc = ctx.nmf_process(X, 7, 9, True, True, 42, progress=lambda i: (seen.append(i), True)[1])
assert seen == list(range(1, 10)), seen
rW, rH, rV, _ = o.nmf_process(X, 7, 9, True, True, 42)
worst = max(worst, rel_err(W1, rW), rel_err(H1, rH))
seen = []
W1, H1, V1, rc = ctx.nmf_process(X, 7, 40, True, True, 42, progress=lambda i: (seen.append(i), i < 5)[1])
assert rc == fluhip.CANCELLED and seen == [1, 2, 3, 4, 5], (rc, seen)
# a corpus of several buffers on the strip schedule
audio = np.stack([oracle_np.synth_audio(30000, 1000 + b) for b in range(5)])
c = fluhip.Corpus(ctx, 5, 30000, 1024, 1024, 256, 8)
assert c.plan()["strip"] == WANT, c.plan()   # 1: fused form + reduce launch, 2: bin strips for W (A/B), 3: bin-tiled W update
c.set_audio(audio); c.stft(); c.nmf(10, seed=42)
mag, W1, H1 = c.read_f64()
for b in (0, 4):
    _, rmag = o.stft_f32(audio[b], 1024, 1024, 256)
    rW, rH, _, _ = o.nmf_process(rmag, 8, 10, True, True, 42)
    e = max(rel_err(W1[b], rW), rel_err(H1[b], rH))
    print("corpus", b, e)
    worst = max(worst, e)
# ... and at fft 2048 (nine bin pairs per wavefront), three buffers, seeds per buffer
audio = np.stack([oracle_np.synth_audio(40000, 2000 + b) for b in range(3)])
c = fluhip.Corpus(ctx, 3, 40000, 2048, 2048, 512, 16)
assert c.plan()["strip"] == WANT, c.plan()   # 1: fused form + reduce launch, 2: bin strips for W (A/B), 3: bin-tiled W update
c.set_audio(audio); c.stft(); c.nmf(6, seeds=[5, 6, 5])
mag, W1, H1 = c.read_f64()
for b in range(3):
    _, rmag = o.stft_f32(audio[b], 2048, 2048, 512)
    rW, rH, _, _ = o.nmf_process(rmag, 16, 6, True, True, [5, 6, 5][b])
    e = max(rel_err(W1[b], rW), rel_err(H1[b], rH))
    print("corpus fft 2048", b, e)
    worst = max(worst, e)
print("worst", worst)
assert worst < 1e-9, worst
'''


@pytest.mark.parametrize("form", ["production", "fused", "bin_strips", "bin_tiles"])
def test_strip_schedule_against_the_oracle(ab_lib_paths, fluhip_lib_path, form):
    """production: the shipped library, no switches.  fused: W partials behind the H phase + the reduce launch (what short buffers and fft 1024 get); bin_strips: the W update
    as its own launch over bin strips with a last-arriver combine (FLUHIP_STRIP_BIN=1, A/B build: measured slower, kept as a
    tested alternative); bin_tiles: the round-5 W update over tiles of four bins and ALL frames (kernels_nmf_bintile.hip;
    FLUHIP_STRIP_TILE=1, A/B build: measured slower at config 2, profiles/r05/c2_bintile.md), forced here onto every shape it supports -- one and a few stages per wavefront, wavefronts without any,
    fft 1024 and 2048, rank 1 .. 16, batches, the update-flag combinations (the Nyquist partials then come from a strip launch
    of their own), seeded factors; shapes it does not support fall back inside the same run"""
    e = dict(os.environ)
    if form == "production":
        # the PRODUCTION library (it reads no switches): every single-buffer shape of the script that the planner gives the
        # schedule by itself -- rank <= 16, F <= 1120, one round of workgroups -- runs the same kernels the A/B build is forced onto
        e["FLUHIP_LIB"] = fluhip_lib_path
        e["STRIP_TEST_PRODUCTION"] = "1"
    else:
        e["FLUHIP_STRIP"] = "1"
        e["FLUHIP_STRIP_BIN"] = "1" if form == "bin_strips" else "0"
        e["FLUHIP_STRIP_TILE"] = "1" if form == "bin_tiles" else "0"
        e["FLUHIP_LIB"] = ab_lib_paths[0]
    p = subprocess.run([sys.executable, "-c", SCRIPT, ROOT], capture_output=True, text=True, timeout=900, env=e)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-1500:]


def test_strip_schedule_is_what_config_2_gets():
    sys.path.insert(0, os.path.join(ROOT, "flucoma-core_amd"))
    import fluhip
    ctx = fluhip.Context(0)
    c = fluhip.Corpus(ctx, 1, 60 * 44100, 2048, 2048, 512, 16)
    assert c.plan()["strip"] == 1, c.plan()     # (the bin-tiled W update of round 5 measured slower: A/B build only)
    c = fluhip.Corpus(ctx, 1, 453932, 1024, 1024, 512, 3)
    assert c.plan()["strip"] == 1, c.plan()     # config 1
    c = fluhip.Corpus(ctx, 128, 10 * 44100, 2048, 2048, 512, 32)
    assert c.plan()["strip"] == 0, c.plan()
