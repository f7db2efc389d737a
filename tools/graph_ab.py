"""us per NMF iteration of the single-buffer BASELINE shapes with the iteration loop enqueued launch by launch or replayed from
a hipGraph (FLUHIP_GRAPH_ITERS=n, api_corpus.hip corpus_iterate_loop): python tools/graph_ab.py [iters]"""
import os
os.environ.setdefault("FLUHIP_AB", "1")   # the build whose experiment switches are live (build.py --ab)
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "flucoma-core_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import fluhip  # noqa: E402
import oracle_np as onp  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ctx = fluhip.Context(0)
shapes = {"c1": (1, 453932, 1024, 512, 3), "c2": (1, 2646000, 2048, 512, 16), "c4x1": (1, 441000, 2048, 512, 32),
          "8ch": (8, 441000, 2048, 512, 32)}
for name, (B, n, fft, hop, K) in shapes.items():
    audio = np.stack([np.resize(onp.synth_audio(min(n, 500000), 1000 + b), n) for b in range(B)])
    c = fluhip.Corpus(ctx, B, n, fft, fft, hop, K)
    c.set_audio(audio); c.stft()
    c.nmf(50, seed=42); ctx.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        c.nmf(iters, seed=42); ctx.synchronize()
        best = min(best, (time.perf_counter() - t0) / iters * 1e6)
    _, W, H = c.read_f64()
    print("%s graph_iters=%s: %.2f us per iteration (checksum %.12g)" % (name, os.environ.get("FLUHIP_GRAPH_ITERS", "0"), best,
                                                                         float(W.sum() + H.sum())), flush=True)
    c.close()
