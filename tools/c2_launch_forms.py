"""Config 2's frame-strip schedule, launch by launch (run under `rocprofv3 --kernel-trace`; tools/r04_probe1.sh): one call
of 3 iterations issues  strip(W partials of H0) . reduce . [strip(H update + W partials) . reduce] x 2 . strip(H update),
i.e. the strip kernel in its three forms -- W phase only, both phases, H phase only -- whose durations price the
alternative schedules of DESIGN section 3 K7 ("two local launches": an H-only launch + a bin-strip W launch)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flucoma-core_amd"))
import fluhip, synth
ctx = fluhip.Context(0)
n = 2646000
base = synth.synth_audio(441000, 1000)
x = np.tile(base, n // len(base) + 1)[:n]
c = fluhip.Corpus(ctx, 1, n, 2048, 2048, 512, 16)
c.set_audio(x[None, :]); c.stft()
c.nmf(20, seed=42); ctx.synchronize()      # warm
for _ in range(5):
    c.nmf(3, seed=42); ctx.synchronize()
print(c.plan())
