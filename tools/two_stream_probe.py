"""Does the bench shard gain from two half-shards running side by side on two streams (512 wavefronts each, out of step) instead of
one launch of 1024 in lock step?  Two contexts (= two streams), one 64-buffer corpus each, driven from two host threads; against
one 128-buffer corpus.  Run with FLUHIP_LIST_PLAN=0 FLUHIP_PLAN_SPLIT=1 (whole contractions, the uniform kernel):
    FLUHIP_LIST_PLAN=0 FLUHIP_PLAN_SPLIT=1 python tools/two_stream_probe.py [iters]"""
import os
os.environ.setdefault("FLUHIP_AB", "1")   # the build whose experiment switches are live (build.py --ab)
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "flucoma-core_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import fluhip  # noqa: E402
import oracle_np as onp  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n, fft, hop, K = 441000, 2048, 512, 32
base = [onp.synth_audio(n, 1000 + b) for b in range(4)]


def corpus(ctx, B):
    c = fluhip.Corpus(ctx, B, n, fft, fft, hop, K)
    c.set_audio(np.stack([base[b % 4] for b in range(B)])); c.stft(); c.nmf(5, seed=42); ctx.synchronize()
    return c


ctxA, ctxB = fluhip.Context(0), fluhip.Context(0)
one = corpus(ctxA, 128)
print("plan of the 128-buffer corpus", one.plan(), flush=True)
for _ in range(3):
    t0 = time.perf_counter(); one.nmf(iters, seed=42); ctxA.synchronize()
    print("one corpus of 128: %.1f us per iteration" % ((time.perf_counter() - t0) / iters * 1e6), flush=True)
one.close()
ca, cb = corpus(ctxA, 64), corpus(ctxB, 64)
print("plan of a 64-buffer corpus", ca.plan(), flush=True)
t0 = time.perf_counter(); ca.nmf(iters, seed=42); ctxA.synchronize()
print("one corpus of 64 alone: %.1f us per iteration" % ((time.perf_counter() - t0) / iters * 1e6), flush=True)


def run(c, ctx):
    c.nmf(iters, seed=42); ctx.synchronize()


for _ in range(3):
    ta, tb = threading.Thread(target=run, args=(ca, ctxA)), threading.Thread(target=run, args=(cb, ctxB))
    t0 = time.perf_counter(); ta.start(); tb.start(); ta.join(); tb.join()
    print("two corpora of 64 side by side: %.1f us per iteration of both" % ((time.perf_counter() - t0) / iters * 1e6), flush=True)
