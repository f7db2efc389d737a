"""end-to-end time of BufNMF with resynthesis (third output buffer): python tools/e2e_resynth.py"""
import sys, time, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flucoma-core_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import fluhip, synth
ctx = fluhip.Context(0)
for name, n, win, fft, hop, K, iters in (("c1", 453932, 1024, 1024, 512, 3, 50), ("c2", 2646000, 2048, 2048, 512, 16, 200)):
    base = synth.synth_audio(min(n, 441000), 1000)
    x = np.tile(base, n // len(base) + 1)[:n].astype(np.float32)
    for rs in (False, True):
        for _ in range(3):   # first calls of a shape load kernels and grow the block cache
            ctx.bufnmf_channel(x, win, fft, hop, K, iters, 42, resynth=rs)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); ctx.bufnmf_channel(x, win, fft, hop, K, iters, 42, resynth=rs); ts.append(time.perf_counter() - t0)
        print(f"{name}: resynth={rs}: {min(ts)*1e3:.2f} ms end to end")
