"""time of the corpus-form resynthesis at the bench shape: python tools/corpus_resynth_timing.py"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flucoma-core_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import fluhip, synth
ctx = fluhip.Context(0)
B, n, K = 128, 441000, 32
base = np.stack([synth.synth_audio(n, 1000 + b) for b in range(4)])
c = fluhip.Corpus(ctx, B, n, 2048, 2048, 512, K)
c.keep_spectrum(True)
c.set_audio(np.tile(base, (B // 4, 1))); c.stft(); c.nmf(10, seed=42); ctx.synchronize()
out = torch.empty((B, K, n), dtype=torch.float32, device="cuda")
import ctypes
for i in range(2):
    t0 = time.perf_counter()
    rc = ctx.lib.fluhip_corpus_resynth_dev(c.h, ctypes.c_void_p(out.data_ptr())); ctx.synchronize()
    print(f"corpus resynthesis {B} x {K} components x {n} samples: {1e3*(time.perf_counter()-t0):.1f} ms (rc {rc})")
print("sum of components vs input, buffer 0:", float((out[0].sum(dim=0).cpu() - torch.from_numpy(base[0])).abs().max()))
