"""Frame-strip schedule (kernels_nmf_strip.hip) on BASELINE config 2's shape: per-launch kernel times by update flags and
the phase stamps of workgroup 0 (FLUHIP_STRIP_INSTR=1: shader-clock and 100 MHz stamps around staging / H phase /
combine / W phase / partial stores).   usage: python tools/strip_timing.py [seconds=60] [rank=16] [fft=2048]"""
import ctypes, os, sys, time
os.environ.setdefault("FLUHIP_AB", "1")   # the build whose experiment switches are live (build.py --ab)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flucoma-core_amd"))
os.environ.setdefault("FLUHIP_STRIP_INSTR", "1")
import fluhip, synth
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
fft = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
n = int(secs * 44100)
ctx = fluhip.Context(0)
c = fluhip.Corpus(ctx, 1, n, fft, fft, fft // 4, K)
print("plan", c.plan(), "frames", c.T, "bins", c.F)
c.set_audio(synth.synth_audio(n, 1000)[None, :]); c.stft()
for (uw, uh, label) in ((True, True, "W+H"), (False, True, "H only"), (True, False, "W only")):
    c.nmf(5, seed=42, updateW=uw, updateH=uh); ctx.synchronize()
    ctx.prof_enable(True); ctx.prof_reset()
    t0 = time.perf_counter()
    c.nmf(50, seed=42, updateW=uw, updateH=uh); ctx.synchronize()
    wall = (time.perf_counter() - t0) / 50 * 1e6
    l1, ms1 = ctx.prof_read(1); l3, ms3 = ctx.prof_read(3)
    ctx.prof_enable(False)
    print(f"{label:7s}: strip launches {l1:4d} x {ms1 / max(l1, 1) * 1e3:7.2f} us   reduce launches {l3:4d} x {ms3 / max(l3, 1) * 1e3:7.2f} us   "
          f"(wall with the per-launch events {wall:7.1f} us per iteration)")
ctx.prof_enable(False)
t0 = time.perf_counter(); c.nmf(200, seed=42); ctx.synchronize()
print(f"200 iterations, no events: {(time.perf_counter() - t0) / 200 * 1e6:7.2f} us per iteration")
out = (ctypes.c_int64 * 32)()
if ctx.lib.fluhip_corpus_debug_words(c.h, out) == 0 and out[0]:
    names = ["prologue (stats, first loads)", "H phase loop", "H reduce + combine", "W phase loop", "tail"]
    cyc = [out[i] for i in range(16)]
    us = (out[16 + 5] - out[16]) / 100.0
    ghz = (cyc[5] - cyc[0]) / us / 1e3
    for i, nme in enumerate(names):
        dc = cyc[i + 1] - cyc[i]
        print(f"  {nme:30s} {dc:8d} cycles  {dc / ghz / 1e3:7.2f} us")
    print(f"  workgroup 0 total              {cyc[5] - cyc[0]:8d} cycles  {us:7.2f} us  ({ghz:5.2f} GHz)")
    print("  prologue detail: to the first request", cyc[10] - cyc[0], " statistics requests", cyc[11] - cyc[10], " rows of W", cyc[12] - cyc[11],
          " H", cyc[13] - cyc[12], " V + rest", cyc[9] - cyc[13], " | column norms known", cyc[15] - cyc[9], " then to the first tile", cyc[1] - cyc[15])
    print("  H phase, third pair of wavefront 0: lane permutation + first product", cyc[7] - cyc[6], " quotients", cyc[8] - cyc[7],
          " second product", cyc[14] - cyc[8])
