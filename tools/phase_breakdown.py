"""Per-phase cycle breakdown of one wavefront of the W-update (FLUHIP_K5_INSTR=1 build path)."""
import ctypes, os, sys
os.environ.setdefault("FLUHIP_AB", "1")   # the build whose experiment switches are live (build.py --ab)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flucoma-core_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ["FLUHIP_K5_INSTR"] = "1"
import fluhip, oracle_np
ctx = fluhip.Context(0)
# default: the bench shard; `python tools/phase_breakdown.py c2` : the single 60 s buffer at rank 16 (split contraction)
if len(sys.argv) > 1 and sys.argv[1] == "c2":
    B, n, K = 1, 2646000, 16
else:
    B, n, K = 128, 441000, 32
x = oracle_np.synth_audio(441000, 1000)
x = np.tile(x, n // len(x) + 1)[:n]
c = fluhip.Corpus(ctx, B, n, 2048, 2048, 512, K)
print("plan", c.plan())
c.set_audio(np.tile(x, (B, 1))); c.stft(); c.nmf(3, seed=42, updateH=False); ctx.synchronize()
out = (ctypes.c_int64 * 32)()
assert ctx.lib.fluhip_corpus_debug_words(c.h, out) == 0
names = ["wait_dma", "lds_read", "ratio", "q_phase", "out_phase", "dma_issue"]
iters = out[6]
tot = sum(out[i] for i in range(6))
print("half-iterations measured:", iters, " (first half of each 2-step loop body)")
for i, nme in enumerate(names):
    print(f"  {nme:10s} {out[i]/iters:8.0f} cycles/step  {100*out[i]/tot:5.1f} %")
print(f"  total      {tot/iters:8.0f} cycles/step")
if out[7]:
    us = out[7] / 100.0   # s_memrealtime counts at 100 MHz
    print(f"  loop wall  {us:8.1f} us  -> shader clock {tot/us/1e3:6.3f} GHz if s_memtime counts shader cycles")

# timeline of wavefront 0 of workgroups 17, 81, 145, 209 (s_memrealtime, 10 ns ticks), relative to the earliest entry
ent = [out[16 + 4 * i] for i in range(4)]
if all(ent):
    t0 = min(ent)
    for i in range(4):
        e, l0, l1, x = (out[16 + 4 * i + j] for j in range(4))
        print(f"  wg {17 + 64 * i:3d}: entry +{(e - t0) / 100:6.2f} us  prologue {(l0 - e) / 100:6.2f} us  loop {(l1 - l0) / 100:7.2f} us  "
              f"epilogue {(x - l1) / 100:6.2f} us  exit +{(x - t0) / 100:7.2f} us")
if out[12] and ent[0]:
    e = out[16]
    print(f"  wg  17 prologue: strip bookkeeping {(out[12]-e)/100:5.2f} us | first 2 stages issued + stationary rows loaded and normalised "
          f"{(out[13]-out[12])/100:5.2f} us | rest of the ring issued, stages 0-1 landed {(out[14]-out[13])/100:5.2f} us | first Q + operand reads {(out[17]-out[14])/100:5.2f} us")
