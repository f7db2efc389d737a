"""Where do replicas of one input differ inside a corpus, and by how much?  (round 5: the random sweep's replica check)
    python tools/probes/r05/replica_diff.py B n win fft hop K iters uw uh"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
for d in ("tests", "flucoma-core_amd", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, d))
import fluhip  # noqa: E402
import oracle_np as onp  # noqa: E402
from helpers import rel_err  # noqa: E402

B, n, win, fft, hop, K, iters, uw, uh = (int(v) for v in sys.argv[1:10])
ctx = fluhip.Context(0, fluhip.load_library(os.path.join(ROOT, "flucoma-core_amd", "lib", "libflucoma_hip.so")))
distinct = [onp.synth_audio(n, 7000 + b) for b in range(min(B, 3))]
audio = np.stack([distinct[b % len(distinct)] for b in range(B)])
c = fluhip.Corpus(ctx, B, n, win, fft, hop, K)
c.set_audio(audio); c.stft()
c.nmf(iters, seed=42, updateW=bool(uw), updateH=bool(uh))
mag, W1, H1 = c.read_f64()
print(c.plan())
c.close()
nd = len(distinct)
bad = []
for b in range(nd, B):
    r = b % nd
    if not (np.array_equal(W1[b], W1[r]) and np.array_equal(H1[b], H1[r])):
        bad.append((b, rel_err(W1[b], W1[r]), rel_err(H1[b], H1[r]), np.array_equal(mag[b], mag[r])))
print(len(bad), "replicas differ from the first copy of their input")
for t in bad[:40]:
    print("  b=%d  W rel %.3e  H rel %.3e  mag equal %s" % t)
