"""STFT kernel time for corpus shapes (fft 1024 and 2048 wave kernels): python tools/stft_timing.py"""
import sys, time, numpy as np, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flucoma-core_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import fluhip, oracle_np
ctx = fluhip.Context(0)
for (B, n, win, fft, hop) in ((2048, 88200, 1024, 1024, 512), (128, 441000, 2048, 2048, 512), (128, 441000, 1024, 1024, 256)):
    base = np.stack([oracle_np.synth_audio(n, 1000 + b) for b in range(4)])
    c = fluhip.Corpus(ctx, B, n, win, fft, hop, 4)
    c.set_audio(np.tile(base, (B // 4, 1)))
    c.stft(); ctx.synchronize()
    ctx.prof_enable(True); ctx.prof_reset()
    for _ in range(5): c.stft()
    nl, ms = ctx.prof_read(0); ctx.prof_enable(False)
    frames = c.T * B
    by = (hop * 4 + c.F * 8) * frames
    print(f"B={B} n={n} fft={fft} hop={hop}: {ms/nl*1e3:.0f} us per launch, {frames/(ms/nl*1e-3)/1e6:.0f} Mframes/s, {by/(ms/nl*1e-3)/1e12:.2f} TB/s algorithmic")
    c.close()
