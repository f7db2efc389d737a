"""How far the device factors are from the oracle at the named iteration counts of the BASELINE configs -- the levels behind the
bars of the tests -- stated three ways: norm-wise (max |a - b| / max |b|, what the 1e-9 bars use),
ELEMENT-wise (max |a - b| / |b| over the entries above 1e-6 of the largest: north_star's "W/H within 1e-5 relative" read entry
by entry) and as a histogram of distances in units of the last place.  Writes gpurun_out/parity_levels.json (copied to
profiles/rNN/); c1 on the reference's own WAV samples (tests/golden/reference_c1.npz)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flucoma-core_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fluhip, oracle_c, oracle_np
from helpers import elementwise_rel_err, rel_err, ulp_histogram
o = oracle_c.get("native")
ctx = fluhip.Context(0)
rec = {"floor": "entries above 1e-6 of the largest entry of the oracle's matrix", "cases": []}


def levels(tag, W1, H1, rW, rH, extra=None):
    e = {"case": tag}
    for name, a, b in (("W", W1, rW), ("H", H1, rH)):
        e[name] = {"normwise": rel_err(a, b), "elementwise": elementwise_rel_err(a, b), **ulp_histogram(a, b)}
    if extra:
        e.update(extra)
    rec["cases"].append(e)
    print(tag, {k: (e[k]["normwise"], e[k]["elementwise"]) for k in ("W", "H")}, flush=True)


# config 4's shard buffer: batched rank-32 schedule, 10 / 50 / 200 iterations
B, n, win, fft, hop, K = 16, 441000, 2048, 2048, 512, 32
audio = np.stack([oracle_np.synth_audio(n, 1000 + (b % 2)) for b in range(B)])
c = fluhip.Corpus(ctx, B, n, win, fft, hop, K)
c.set_audio(audio); c.stft()
_, rmag = o.stft_f32(audio[0], win, fft, hop)
for iters in (10, 50, 200):
    c.nmf(iters, seed=42)
    mag, W1, H1 = c.read_f64()
    rW, rH, _, _ = o.nmf_process(rmag, K, iters, True, True, 42)
    levels(f"c4 shard buffer, rank 32, {iters} iterations", W1[0], H1[0], rW, rH)
c.close()
# config 2: 60 s, rank 16, the frame-strip schedule, all 200 iterations
n = 2646000
x = np.tile(oracle_np.synth_audio(441000, 1000), 6)[:n]
c = fluhip.Corpus(ctx, 1, n, win, fft, hop, 16)
c.set_audio(x[None, :]); c.stft(); c.nmf(200, seed=42)
mag, W1, H1 = c.read_f64()
c.close()
_, rmag = o.stft_f32(x, win, fft, hop)
rW, rH, _, _ = o.nmf_process(rmag, 16, 200, True, True, 42)
levels("c2: 60 s, rank 16, 200 iterations", W1[0], H1[0], rW, rH)
# config 1 on the reference's own samples: rank 3, fft 1024 / hop 512, 50 iterations
g = np.load(os.path.join(ROOT, "tests", "golden", "reference_c1.npz"))
win1, fft1, hop1, K1, it1, seed1 = (int(v) for v in g["params"])
x = g["pcm16"].astype(np.float32) / 32768.0
c = fluhip.Corpus(ctx, 1, len(x), win1, fft1, hop1, K1)
c.set_audio(x[None, :]); c.stft(); c.nmf(it1, seed=seed1)
mag, W1, H1 = c.read_f64()
c.close()
_, rmag = o.stft_f32(x, win1, fft1, hop1)
rW, rH, _, _ = o.nmf_process(rmag, K1, it1, True, True, seed1)
levels("c1: Nicol-LoopE-M.wav, rank 3, 50 iterations", W1[0], H1[0], rW, rH)
# config 3's 12 s twin: fft 4096 / hop 1024, rank 128, all 500 iterations
n = 529200
x = oracle_np.synth_audio(n, 1000)
c = fluhip.Corpus(ctx, 1, n, 4096, 4096, 1024, 128)
c.set_audio(x[None, :]); c.stft(); c.nmf(500, seed=42)
mag, W1, H1 = c.read_f64()
c.close()
rW, rH, _, _ = o.nmf_process(mag[0], 128, 500, True, True, 42)
levels("c3 twin: 12 s, fft 4096, rank 128, 500 iterations", W1[0], H1[0], rW, rH)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rec, open(os.path.join(ROOT, "gpurun_out", "parity_levels.json"), "w"), indent=1)
