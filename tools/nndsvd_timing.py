import sys, time, numpy as np
sys.path.insert(0, "flucoma-core_amd"); sys.path.insert(0, "oracle")
import fluhip
ctx = fluhip.Context(0)
rs = np.random.RandomState(0)
for (T, F) in ((60, 33), (60, 33), (200, 129), (300, 513), (887, 513), (5168, 1025)):
    X = np.abs(rs.standard_normal((T, 12))) @ np.abs(rs.standard_normal((12, F))) + 1e-3
    t0 = time.perf_counter(); W, H, k = ctx.nndsvd(X, 16, 1, 16, 0.8, 0, 42); dt = time.perf_counter() - t0
    t0 = time.perf_counter(); np.linalg.svd(X.T, full_matrices=False); dn = time.perf_counter() - t0
    print(f"T={T} F={F}: fluhip_nndsvd {dt*1e3:.1f} ms (rank {k}); numpy/LAPACK svd {dn*1e3:.1f} ms", flush=True)
