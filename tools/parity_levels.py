"""How far the device factors are from the oracle after 10 / 50 / 200 iterations (the actual levels behind the 1e-9 bars of the\ntests): the batched rank-32 schedule of the bench shard and the frame-strip schedule at rank 16."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flucoma-core_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fluhip, oracle_c, oracle_np
from helpers import rel_err
o = oracle_c.get("native")
ctx = fluhip.Context(0)
B, n, win, fft, hop, K = 16, 441000, 2048, 2048, 512, 32
audio = np.stack([oracle_np.synth_audio(n, 1000 + (b % 2)) for b in range(B)])
c = fluhip.Corpus(ctx, B, n, win, fft, hop, K)
c.set_audio(audio); c.stft()
_, rmag = o.stft_f32(audio[0], win, fft, hop)
for iters in (10, 50, 200):
    c.nmf(iters, seed=42)
    mag, W1, H1 = c.read_f64()
    rW, rH, _, _ = o.nmf_process(rmag, K, iters, True, True, 42)
    print("rank 32 batched", iters, "iterations: W", rel_err(W1[0], rW), "H", rel_err(H1[0], rH), flush=True)
c.close()
n = 10 * 44100
a = oracle_np.synth_audio(n, 1000)
c = fluhip.Corpus(ctx, 1, n, win, fft, hop, 16)
c.set_audio(a[None, :]); c.stft()
_, rmag = o.stft_f32(a, win, fft, hop)
for iters in (10, 50, 200):
    c.nmf(iters, seed=42)
    mag, W1, H1 = c.read_f64()
    rW, rH, _, _ = o.nmf_process(rmag, 16, iters, True, True, 42)
    print("rank 16 strip", iters, "iterations: W", rel_err(W1[0], rW), "H", rel_err(H1[0], rH), flush=True)
