"""Per-iteration time of the frame-strip schedule (FLUHIP_STRIP=1) against the split-contraction schedule (FLUHIP_STRIP=0) for
single buffers of several lengths / fft sizes / ranks <= 16: where the planner's threshold comes from.
usage: python tools/strip_vs_split.py"""
import os, subprocess, sys
os.environ.setdefault("FLUHIP_AB", "1")   # the build whose experiment switches are live (build.py --ab)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(sys.argv[1], "flucoma-core_amd"))
import fluhip, synth
ctx = fluhip.Context(0)
for spec in sys.argv[2:]:
    secs, fft, K, B = spec.split(":")
    secs, fft, K, B = float(secs), int(fft), int(K), int(B)
    n = int(secs * 44100)
    c = fluhip.Corpus(ctx, B, n, fft, fft, fft // 4, K)
    c.set_audio(np.tile(synth.synth_audio(n, 1000)[None, :], (B, 1))); c.stft()
    c.nmf(10, seed=42); ctx.synchronize()
    t0 = time.perf_counter(); c.nmf(100, seed=42); ctx.synchronize()
    print(spec, c.T, c.plan()["strip"], (time.perf_counter() - t0) / 100 * 1e6)
'''
specs = sys.argv[1:] or ["60:2048:16:1", "30:2048:16:1", "10:2048:16:1", "5:2048:16:1", "2:2048:16:1", "60:1024:16:1", "10:1024:16:1",
                         "3:1024:3:1", "120:2048:8:1", "600:2048:16:1", "60:2048:16:2", "10:2048:16:4", "10:512:12:1"]
res = {}
for mode in ("0", "1"):
    e = dict(os.environ); e["FLUHIP_STRIP"] = mode
    p = subprocess.run([sys.executable, "-c", CHILD, ROOT] + specs, capture_output=True, text=True, env=e, timeout=600)
    if p.returncode: print(p.stderr[-2000:])
    for line in p.stdout.splitlines():
        spec, T, strip, us = line.split()
        res.setdefault(spec, {})[mode] = (int(T), int(strip), float(us))
print(f"{'seconds:fft:rank:buffers':26s} {'frames':>7s} {'split us/it':>12s} {'strip us/it':>12s}  ratio")
for spec in specs:
    r = res.get(spec, {})
    if "0" in r and "1" in r:
        print(f"{spec:26s} {r['0'][0]:7d} {r['0'][2]:12.1f} {r['1'][2]:12.1f}  {r['0'][2] / r['1'][2]:5.2f}" + ("" if r["1"][1] else "  (strip not taken)"))
