"""STFT phase of a corpus, timed by the library's own event pairs (fluhip_prof_*): us per launch of the block kernel.
    [FLUHIP_AB=1 FLUHIP_STFT_DB1K=1] python tools/stft_timing.py fft [buffers [seconds [hop]]]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flucoma-core_amd"))
import fluhip  # noqa: E402

fft = int(sys.argv[1])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
seconds = float(sys.argv[3]) if len(sys.argv) > 3 else 10.0
hop = int(sys.argv[4]) if len(sys.argv) > 4 else fft // 4
n = int(seconds * 44100)
ctx = fluhip.Context()
rs = np.random.RandomState(1)
audio = (rs.rand(B, n).astype(np.float32) - 0.5)
c = fluhip.Corpus(ctx, B, n, fft, fft, hop, 8)
c.set_audio(audio)
c.stft()
ctx.prof_enable(True)
ctx.prof_reset()
for _ in range(20):
    c.stft()
cnt, ms = ctx.prof_read(0)
mag = c.read_f64(factors=False)[0]
print("fft %d, %d x %.0f s, hop %d: %.1f us per STFT phase (%d timed), checksum %.9g" % (fft, B, seconds, hop, ms / max(cnt, 1) * 1e3, cnt, float(mag.sum())))
c.close()
