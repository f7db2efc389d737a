"""A ragged corpus against its equal-length twin (same number of buffers, same total frames):
    python tools/ragged_timing.py [buffers] [distinct lengths] [min s] [max s] [K] [which: both|ragged|equal]
per-iteration wall time, the schedules, HIP-event sums per kernel class.  Under rocprofv3 --kernel-trace --stats the
work-list instantiations of the update kernel show up under their own names (last template argument 1)."""
import sys, os, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flucoma-core_amd"))
import fluhip, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nd = int(sys.argv[2]) if len(sys.argv) > 2 else 40
lo = float(sys.argv[3]) if len(sys.argv) > 3 else 4.0
hi = float(sys.argv[4]) if len(sys.argv) > 4 else 16.0
K = int(sys.argv[5]) if len(sys.argv) > 5 else 32
which = sys.argv[6] if len(sys.argv) > 6 else "both"
win, fft, hop = 2048, 2048, 512
rs = np.random.RandomState(7)
distinct = sorted(int(x) for x in rs.randint(int(lo * 44100), int(hi * 44100), nd))
lens = [distinct[i % nd] for i in range(B)]
rs.shuffle(lens)
base = [synth.synth_audio(int(hi * 44100), 5000 + i) for i in range(8)]
ctx = fluhip.Context(0)


def rate(c, n_it=60):
    c.nmf(10, seed=42); ctx.synchronize()
    t0 = time.perf_counter(); c.nmf(0, seed=42); ctx.synchronize(); t_fixed = time.perf_counter() - t0
    t0 = time.perf_counter(); c.nmf(n_it, seed=42); ctx.synchronize()
    t = (time.perf_counter() - t0 - t_fixed) / n_it
    ctx.prof_enable(True); ctx.prof_reset()
    c.nmf(n_it, seed=42); ctx.synchronize()
    k = {"updates_ms_per_iteration": ctx.prof_read(1)[1] / n_it, "between_ms_per_iteration": ctx.prof_read(3)[1] / n_it}
    ctx.prof_enable(False)
    return t, k


out = {"buffers": B, "distinct_lengths": nd, "frames_total": sum((n + hop) // hop for n in lens)}
if which in ("both", "ragged"):
    c = fluhip.RaggedCorpus(ctx, lens, win, fft, hop, K)
    c.set_audio([base[i % 8][:n] for i, n in enumerate(lens)]); c.stft()
    t, k = rate(c)
    out["ragged"] = {"us_per_iteration": t * 1e6, "plan": c.plan(), **k}
    c.close()
if which in ("both", "equal"):
    n_eq = int(sum(lens) / len(lens))
    u = fluhip.Corpus(ctx, B, n_eq, win, fft, hop, K)
    u.set_audio(np.stack([base[i % 8][:n_eq] for i in range(B)])); u.stft()
    t, k = rate(u)
    out["equal"] = {"us_per_iteration": t * 1e6, "plan": u.plan(), **k}
    u.close()
print(json.dumps(out))
