"""STFT phase time for corpus shapes (both magnitude layouts written): python tools/stft_timing.py
FLUHIP_STFT_BLOCK selects the block-kernel variant (0 = the round-1 wave kernel + transposing copy)."""
import sys, time, numpy as np, os
os.environ.setdefault("FLUHIP_AB", "1")   # the build whose experiment switches are live (build.py --ab)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flucoma-core_amd"))
import fluhip, synth
ctx = fluhip.Context(0)
SHAPES = ((128, 441000, 2048, 2048, 512), (2048, 88200, 1024, 1024, 512), (128, 441000, 1024, 1024, 256), (4, 26460000, 4096, 4096, 1024))
sel = os.environ.get('STFT_SHAPES')
for (B, n, win, fft, hop) in (SHAPES if sel is None else [SHAPES[int(i)] for i in sel.split(',')]):
    base = np.stack([synth.synth_audio(n, 1000 + b) for b in range(4)]) if n < 4000000 else np.stack([np.tile(synth.synth_audio(441000, 1000 + b), n // 441000) for b in range(4)])
    c = fluhip.Corpus(ctx, B, n, win, fft, hop, 4)
    c.set_audio(np.tile(base, (B // 4, 1)))
    c.stft(); ctx.synchronize()
    ctx.prof_enable(True); ctx.prof_reset()
    for _ in range(5): c.stft()
    nl, ms = ctx.prof_read(0)
    nt, mt = ctx.prof_read(4)
    ctx.prof_enable(False)
    t0 = time.perf_counter()
    for _ in range(5): c.stft()
    ctx.synchronize()
    wall = (time.perf_counter() - t0) / 5
    frames = c.T * B
    by = (hop * 4 + c.F * 8) * frames
    print(f"[{os.environ.get('FLUHIP_STFT_BLOCK', 'default')}] B={B} n={n} fft={fft} hop={hop}: stft kernel {ms/nl*1e3:.0f} us, transpose {mt/max(nt,1)*1e3:.0f} us, "
          f"phase wall {wall*1e6:.0f} us -> {frames/wall/1e6:.0f} Mframes/s, {by/wall/1e12:.2f} TB/s algorithmic (kernel alone {frames/(ms/nl*1e-3)/1e6:.0f} Mframes/s)", flush=True)
    c.close()
