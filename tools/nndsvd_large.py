import sys, time, numpy as np
sys.path.insert(0, "flucoma-core_amd"); sys.path.insert(0, "oracle")
import fluhip, oracle_np as onp
ctx = fluhip.Context(0)
rs = np.random.RandomState(1)
for (T, F, r) in ((3000, 2049, 40), (25840, 2049, 60)):
    X = (np.abs(rs.standard_normal((T, r))) * np.linspace(3, 0.2, r)) @ np.abs(rs.standard_normal((r, F))) + 1e-3 * rs.uniform(0, 1, (T, F))
    t0 = time.perf_counter(); W, H, k = ctx.nndsvd(X, 32, 1, 32, 0.9, 0, 42); dt = time.perf_counter() - t0
    if T <= 3000:
        t0 = time.perf_counter(); rW, rH, rk, *_ = onp.nndsvd(X, 32, 1, 32, 0.9, 0, 42); dn = time.perf_counter() - t0
        print(f"T={T} F={F}: gpu {dt:.2f} s rank {k}; numpy {dn:.2f} s rank {rk}; relerr W {np.abs(W-rW).max()/np.abs(rW).max():.2e} H {np.abs(H-rH).max()/np.abs(rH).max():.2e}", flush=True)
    else:
        print(f"T={T} F={F}: gpu {dt:.2f} s rank {k}", flush=True)
