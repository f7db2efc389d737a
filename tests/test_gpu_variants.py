"""The kernel forms that production does not pick on its own (environment switches read once per process, so each runs in
a subprocess): the un-fused any-rank factor-update path (FLUHIP_NMF_KERNEL=-1), the update kernel's other pipeline forms
and schedules -- the same shapes against the oracle, so the A/B switches of DESIGN section 6b stay trustworthy;
plus the STFT forms (round-1 wave kernel + transposing copy, generic workgroup-per-frame kernel) and the split-contraction
schedule of single buffers at rank <= 16 (FLUHIP_STRIP=0; the frame-strip schedule is what they get by default)."""
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
for (T, F, K, iters) in ((200, 1025, 16, 12), (173, 1025, 32, 12), (97, 257, 48, 8), (64, 129, 64, 8), (33, 17, 1, 10), (300, 513, 3, 10)):
    rs = np.random.RandomState(T * 7 + F)
    X = np.abs(rs.standard_normal((T, 3)) @ rs.standard_normal((3, F))) + 0.01 * rs.uniform(0, 1, (T, F))
    W1, H1, V1, rc = ctx.nmf_process(X, K, iters, True, True, 42)
    rW, rH, rV, _ = o.nmf_process(X, K, iters, True, True, 42)
    worst = max(worst, rel_err(W1, rW), rel_err(H1, rH), rel_err(V1, rV))
# a corpus (batched schedule) and the single-buffer split schedule
audio = np.stack([oracle_np.synth_audio(30000, 1000 + b) for b in range(9)])
c = fluhip.Corpus(ctx, 9, 30000, 1024, 1024, 256, 8)
c.set_audio(audio); c.stft(); c.nmf(10, seed=42)
mag, W1, H1 = c.read_f64()
for b in (0, 8):
    _, rmag = o.stft_f32(audio[b], 1024, 1024, 256)
    rW, rH, _, _ = o.nmf_process(rmag, 8, 10, True, True, 42)
    worst = max(worst, rel_err(mag[b], rmag) * 1e3, rel_err(W1[b], rW), rel_err(H1[b], rH))
print("plan", c.plan()["kernel"], "worst", worst)
assert worst < 1e-9, worst
# the batched regime with the Nyquist side column (FLUHIP_SIDE_SLICES=2: its slices then go through the block in chunks,
# as on buffers of more than 256 x 64 frames), ranks 32 and 128
for K in (12, 32, 64, 128):
    audio = np.stack([oracle_np.synth_audio(70000, 1100 + (b % 3)) for b in range(128)])
    c = fluhip.Corpus(ctx, 128, 70000, 2048, 2048, 512, K)
    c.set_audio(audio); c.stft(); c.nmf(4, seed=42)
    mag, W1, H1 = c.read_f64()
    assert c.plan()["side_column"] == 1 or os.environ.get("FLUHIP_NO_SIDE") or os.environ.get("FLUHIP_NMF_KERNEL") or os.environ.get("FLUHIP_NO_LAZY"), c.plan()
    for b in range(3, 128):   # every replica of the three inputs, bit for bit (a fault in a few lanes of a few buffers shows here)
        assert np.array_equal(W1[b], W1[b % 3]) and np.array_equal(H1[b], H1[b % 3]), (K, b)
    for b in (1, 127):
        _, rmag = o.stft_f32(audio[b], 2048, 2048, 512)
        rW, rH, _, _ = o.nmf_process(rmag, K, 4, True, True, 42)
        assert max(rel_err(W1[b], rW), rel_err(H1[b], rH)) < 1e-9, (K, b)
    c.close()
# one long-ish buffer at ranks 112 / 128 (the forms that take their column sums from a pre-pass, the contraction split, 256
# statistics records): the side column in front of the W update with its denominators as the column sums of H, the
# pre-reduced norm combine that also leaves the column sums of W' for the H update (round 5) -- and, with
# FLUHIP_COLSUM_FROM_SIDE=0 / FLUHIP_WNORM_PRE=0 / FLUHIP_FIN_BATCH, the forms they replaced
xl = oracle_np.synth_audio(200000, 4321)
_, magl = o.stft_f32(xl, 2048, 2048, 512)
for K in (112, 128):
    c = fluhip.Corpus(ctx, 1, 200000, 2048, 2048, 512, K)
    c.set_audio(xl[None, :]); c.stft(); c.nmf(6, seed=42)
    _, W1, H1 = c.read_f64()
    rW, rH, _, _ = o.nmf_process(magl, K, 6, True, True, 42)
    assert max(rel_err(W1[0], rW), rel_err(H1[0], rH)) < 1e-9, (K, rel_err(W1[0], rW), rel_err(H1[0], rH))
    c.close()
# resynthesis at fft 2048 against the oracle (FLUHIP_RESYNTH_BATCH=0: the per-buffer frame + overlap-add kernels there;
# FLUHIP_STFT_PREFETCH=0: the STFT's round-2 load order)
x = oracle_np.synth_audio(30000, 77)
bases, acts, res, rc = ctx.bufnmf_channel(x, 2048, 2048, 512, 3, 8, 42, resynth=True)
spec, mag1 = o.stft_f32(x, 2048, 2048, 512)
rW, rH, rV, _ = o.nmf_process(mag1, 3, 8, True, True, 42)
for k in range(3):
    ref = o.resynth_component(spec, rW, rH, rV, k, 2048, 2048, 512, 30000)
    assert np.abs(res[k] - ref).max() / max(np.abs(ref).max(), 1e-12) < 1e-5, k
# (rank 9: two workgroups of eight wavefronts per run, the second with one live component -- the shared-row form)
bases, acts, res, rc = ctx.bufnmf_channel(x, 2048, 2048, 512, 9, 4, 42, resynth=True)
rW, rH, rV, _ = o.nmf_process(mag1, 9, 4, True, True, 42)
for k in range(9):
    ref = o.resynth_component(spec, rW, rH, rV, k, 2048, 2048, 512, 30000)
    assert np.abs(res[k] - ref).max() / max(np.abs(ref).max(), 1e-12) < 1e-5, k
'''


@pytest.mark.parametrize("env", [{"FLUHIP_NMF_KERNEL": "-1"},
                                 {"FLUHIP_NO_LAZY": "1"}, {"FLUHIP_NO_SIDE": "1"}, {"FLUHIP_SIDE_FUSED": "1"}, {"FLUHIP_LIST_PLAN": "0"}, {"FLUHIP_LIST_PLAN": "1"}, {"FLUHIP_STFT_BLOCK": "0"},
                                 {"FLUHIP_STFT_GENERIC": "1"}, {"FLUHIP_K5_MODE": "0"}, {"FLUHIP_K5_MODE": "1"}, {"FLUHIP_TAIL_SPLIT": "0"}, {"FLUHIP_GRAPH_ITERS": "4"}, {"FLUHIP_K5_MODE": "2"}, {"FLUHIP_K5_MODE": "2", "FLUHIP_K5_MODE_ANY": "1"}, {"FLUHIP_STRIP": "0"}, {"FLUHIP_STRIP_BIN": "1"}, {"FLUHIP_STRIP_SIDE": "0"},
                                 {"FLUHIP_RESYNTH_BATCH": "0"}, {"FLUHIP_RESYNTH_SHARED": "0"}, {"FLUHIP_STFT_PREFETCH": "0"},
                                 {"FLUHIP_SIDE_SLICES": "2"}, {"FLUHIP_SIDE_STREAM": "1"}, {"FLUHIP_SIDE_NORM": "0"}, {"FLUHIP_SIDE_FROM_H": "0"}, {"FLUHIP_NORM_IN_H": "0"},
                                 {"FLUHIP_COLSUM_FROM_SIDE": "0"}, {"FLUHIP_WNORM_PRE": "0"}, {"FLUHIP_FIN_BATCH": "8"}, {"FLUHIP_FIN_BATCH": "16"},
                                 {"FLUHIP_SIDE_FIRST_CORPORA": "1"}, {"FLUHIP_STFT_NW": "16"}, {"FLUHIP_SIDE_ROWS": "0"}],
                         ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()))
def test_alternative_kernel_forms_against_the_oracle(env, ab_lib_paths):
    e = dict(os.environ)
    e.update(env)
    e["FLUHIP_LIB"] = ab_lib_paths[0]        # the build that reads the switches; the production library ignores them
    p = subprocess.run([sys.executable, "-c", SCRIPT, ROOT], capture_output=True, text=True, timeout=600, env=e)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    assert ("plan 0 " if env.get("FLUHIP_NMF_KERNEL") == "-1" else "plan 5 ") in p.stdout, p.stdout


def test_the_two_quotients_after_500_iterations_at_rank_128(ab_lib_paths, fluhip_lib_path):
    """The factor updates take V / max(WH, eps) with a Newton-refined reciprocal (relative error <= 2^-46, kernels_nmf5.hip
    `quotient`); -DFLUHIP_QUOTIENT_CORRECTION=1 adds the residual step that makes it a rounding-level quotient.  Config 3's
    regime -- rank 128, all 500 iterations, 12 s twin of the 10-minute channel -- with BOTH builds against the oracle
    (<= 1e-9 each, the bar of every other parity test) and against each other; the distances go to
    gpurun_out/quotient_500it_rank128.json (copied to profiles/ per round)."""
    import json
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "flucoma-core_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fluhip
    import oracle_c
    import oracle_np
    from helpers import rel_err
    n, win, fft, hop, K, iters = 529200, 4096, 4096, 1024, 128, 500
    x = oracle_np.synth_audio(n, 1000)
    got = {}
    for tag, path in (("default", fluhip_lib_path), ("corrected", ab_lib_paths[1])):
        lib = fluhip.load_library(path)
        c0 = fluhip.Context(0, lib)
        c = fluhip.Corpus(c0, 1, n, win, fft, hop, K)
        c.set_audio(x[None, :]); c.stft(); c.nmf(iters, seed=42)
        mag, W1, H1 = c.read_f64()
        got[tag] = (W1[0].copy(), H1[0].copy())
        c.close(); c0.close()
    o = oracle_c.get("native")
    rW, rH, _, _ = o.nmf_process(mag[0], K, iters, True, True, 42)
    rec = {"shape": {"frames": int(mag.shape[1]), "bins": int(mag.shape[2]), "rank": K, "iterations": iters}}
    for tag, (W1, H1) in got.items():
        rec[tag + "_vs_oracle"] = {"W": rel_err(W1, rW), "H": rel_err(H1, rH)}
        assert rel_err(W1, rW) < 1e-9 and rel_err(H1, rH) < 1e-9, (tag, rec)
    rec["default_vs_corrected"] = {"W": rel_err(got["default"][0], got["corrected"][0]),
                                   "H": rel_err(got["default"][1], got["corrected"][1])}
    print(json.dumps(rec))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "quotient_500it_rank128.json"), "w") as f:
        json.dump(rec, f, indent=1)
    assert rec["default_vs_corrected"]["W"] < 1e-9 and rec["default_vs_corrected"]["H"] < 1e-9
