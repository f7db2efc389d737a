import sys, time, numpy as np
sys.path.insert(0, "flucoma-core_amd"); sys.path.insert(0, "oracle")
import fluhip, synth
ctx = fluhip.Context(0)
for name, n, win, fft, hop, K, iters in (("c1", 453932, 1024, 1024, 512, 3, 50), ("c2", 2646000, 2048, 2048, 512, 16, 200), ("c4x1", 441000, 2048, 2048, 512, 32, 200)):
    base = synth.synth_audio(min(n, 441000), 1000)
    x = np.tile(base, n // len(base) + 1)[:n].astype(np.float32)
    ctx.bufnmf_channel(x, win, fft, hop, K, iters, 42)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); ctx.bufnmf_channel(x, win, fft, hop, K, iters, 42); ts.append(time.perf_counter() - t0)
    print(f"{name}: fluhip_bufnmf_channel_f32 end to end (host buffers in/out) {min(ts)*1e3:.2f} ms (median {sorted(ts)[2]*1e3:.2f})")
