"""Bin-strip W launch of the frame-strip schedule (kernels_nmf_strip.hip nmf_binstrip_kernel) on config 2's shape: 100 MHz
stamps of workgroup 0 and of strip 0's last arriver (FLUHIP_STRIP_INSTR=1, A/B build).   python tools/binstrip_timing.py"""
import ctypes, os, sys, time
os.environ.setdefault("FLUHIP_AB", "1")
os.environ.setdefault("FLUHIP_STRIP_INSTR", "1")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flucoma-core_amd"))
import fluhip, synth
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
n = int(secs * 44100)
ctx = fluhip.Context(0)
c = fluhip.Corpus(ctx, 1, n, 2048, 2048, 512, 16)
print("plan", c.plan(), "frames", c.T, "bins", c.F)
c.set_audio(np.resize(synth.synth_audio(441000, 1000), n)[None, :]); c.stft()
c.nmf(20, seed=42); ctx.synchronize()
c.nmf(1, seed=42, updateH=False); ctx.synchronize()
out = (ctypes.c_int64 * 32)()
assert ctx.lib.fluhip_corpus_debug_words(c.h, out) == 0
t = [out[i] / 100.0 for i in range(16)]
names = ["prologue (records, rows of W', norms)", "frame loop", "wavefronts added up, partial stores issued", "stores retired + barrier", "ticket"]
for i, nm in enumerate(names):
    print(f"  workgroup 0: {nm:45s} {t[i + 1] - t[i]:7.2f} us")
print(f"  workgroup 0 total {t[5] - t[0]:7.2f} us")
print(f"  strip 0's last arriver: enters {t[8] - t[0]:7.2f} us after workgroup 0 started; row sums {t[9] - t[8]:6.2f}, numerators {t[10] - t[9]:6.2f}, "
      f"update + statistics {t[11] - t[10]:6.2f}; done at {t[11] - t[0]:7.2f} us")
