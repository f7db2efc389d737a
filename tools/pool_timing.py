"""A whole corpus through the device pool from HOST buffers (fluhip_pool_bufnmf_f32: upload, STFT, NMF, write-back):
    python tools/pool_timing.py [buffers] [iters]
wall time of the job; FLUHIP_POOL_SLICES=0 runs a device's share as one corpus (upload, then compute) instead of slices whose
uploads overlap the previous slice's iterations."""
import sys, os, time, json
os.environ.setdefault("FLUHIP_AB", "1")   # the build whose experiment switches are live (build.py --ab)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flucoma-core_amd"))
import fluhip, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n = 441000
base = np.stack([synth.synth_audio(n, 1000 + b) for b in range(8)])
audio = np.ascontiguousarray(np.tile(base, (B // 8, 1)))
pool = fluhip.Pool([0])
pool.bufnmf(audio[:8], 2048, 2048, 512, 32, 2)
t0 = time.perf_counter()
bases, acts, rc = pool.bufnmf(audio, 2048, 2048, 512, 32, iters)
dt = time.perf_counter() - t0
print(json.dumps({"buffers": B, "iterations": iters, "slices": os.environ.get("FLUHIP_POOL_SLICES", "default"), "wall_ms": dt * 1e3,
                  "buffer_iterations_per_s": B * iters / dt,
                  "checksum": float(bases.astype(np.float64).sum() + acts.astype(np.float64).sum())}))
pool.close()
