"""Where the time of a small batch goes (the channels of one multichannel BufNMF job as one corpus):
    python tools/batch_timing.py [B] [seconds] [K] [iters] [fft] [hop]
wall times of upload / STFT / NMF (with and without a progress callback) / write-back, the schedule the corpus got and the
per-class kernel times of the iteration loop."""
import sys, os, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flucoma-core_amd"))
import fluhip, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
K = int(sys.argv[3]) if len(sys.argv) > 3 else 32
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 200
fft = int(sys.argv[5]) if len(sys.argv) > 5 else 2048
hop = int(sys.argv[6]) if len(sys.argv) > 6 else fft // 4
n = int(secs * 44100)
ctx = fluhip.Context(0)
audio = np.stack([synth.synth_audio(n, 800 + b) for b in range(B)])
c = fluhip.Corpus(ctx, B, n, fft, fft, hop, K)


def timed(f):
    ctx.synchronize(); t0 = time.perf_counter(); f(); ctx.synchronize(); return (time.perf_counter() - t0) * 1e3


c.set_audio(audio); c.stft(); c.nmf(5, seed=42); c.writeback()
res = {"B": B, "n": n, "K": K, "iters": iters, "plan": c.plan()}
res["upload_ms"] = timed(lambda: c.set_audio(audio))
res["stft_ms"] = timed(lambda: c.stft())
res["nmf0_ms"] = timed(lambda: c.nmf(0, seed=42))
res["nmf_ms"] = timed(lambda: c.nmf(iters, seed=42))
res["nmf_progress_ms"] = timed(lambda: c.nmf(iters, seed=42, progress=lambda i: True))
res["writeback_ms"] = timed(lambda: c.writeback())
ctx.prof_enable(True); ctx.prof_reset()
c.nmf(iters, seed=42); ctx.synchronize()
res["kernel_ms"] = {"updates": ctx.prof_read(1), "between": ctx.prof_read(3)}
ctx.prof_enable(False)
res["us_per_iteration"] = (res["nmf_ms"] - res["nmf0_ms"]) / iters * 1e3
res["us_per_iteration_progress"] = (res["nmf_progress_ms"] - res["nmf0_ms"]) / iters * 1e3
print(json.dumps(res))
