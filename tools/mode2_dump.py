"""One W update of a one-step corpus (4 frames) with the in-place pipeline form forced onto rank 32: the numerators the kernel
must have formed, element by element, next to what it wrote (bisecting aid; FLUHIP_LIB / FLUHIP_K5_MODE as mode2_bisect.py)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flucoma-core_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import fluhip, oracle_np
ctx = fluhip.Context(0)
B, n, K = 128, int(sys.argv[1]) if len(sys.argv) > 1 else 1536, 32
audio = np.stack([oracle_np.synth_audio(n, 1100 + (b % 4)) for b in range(B)])
c = fluhip.Corpus(ctx, B, n, 2048, 2048, 512, K)
c.set_audio(audio); c.stft(); c.nmf(0, seed=42)
mag, W0, H0 = c.read_f64()
c.nmf(1, seed=42, updateW=True, updateH=False)
_, W1, H1 = c.read_f64(mag=False)
if os.environ.get("M2_WORDS"):
    import ctypes
    w = (ctypes.c_int64 * 32)()
    assert ctx.lib.fluhip_corpus_debug_words(c.h, w) == 0
    d = np.frombuffer(bytes(w), dtype=np.float64).reshape(8, 4)
    print("lane 3 of wavefront 0, workgroup 0, group 0: m, so, acc, dy, r")
    for m in range(8): print("  ", m, d[m].tolist())
c.close()
eps = 2.220446049250313e-16
out = []
for b in (0, 1, 5):
    V = mag[b].T                      # F x T
    W = W0[b].T                       # F x K (normalised)
    H = H0[b].T                       # K x T
    P = np.maximum(W @ H, eps)
    num = (V / P) @ H.T               # F x K
    den = np.maximum(H.sum(axis=1), eps)
    Wn = W * num / den
    nrm = np.sqrt((Wn ** 2).sum(axis=0))
    Wref = Wn / nrm
    got = W1[b].T
    # undo the normalisation with the reference's norms (the bad entries are few): numerator the kernel must have used
    num_got = got * nrm * den / np.maximum(W, 1e-300)
    rows = [0, 1, 2, 3, 4, 128, 129]
    rec = {"buffer": b, "T": int(V.shape[1])}
    for r in rows:
        rec[f"row{r}"] = {"num_ref": num[r, 20:32].tolist(), "num_got": num_got[r, 20:32].tolist(),
                          "W": W[r, 20:32].tolist(), "V": V[r, :8].tolist()}
    rec["H_comps24_31"] = H[24:32, :8].tolist()
    rec["worst"] = float(np.abs(got - Wref).max() / np.abs(Wref).max())
    bad = np.argwhere(np.abs(got - Wref) > 1e-9 * np.abs(Wref).max())
    rec["bad_rows"] = sorted(set(int(x) for x in bad[:, 0]))[:40]
    rec["bad_comps"] = sorted(set(int(x) for x in bad[:, 1]))
    out.append(rec)
os.makedirs(os.path.join(ROOT, "gpurun_out", "mode2"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "mode2", f"dump_{n}.json"), "w"), indent=1)
for rec in out:
    print(rec["buffer"], rec["worst"], rec["bad_rows"], rec["bad_comps"])
    for r in (0, 1, 128):
        a, g = np.array(rec[f"row{r}"]["num_ref"]), np.array(rec[f"row{r}"]["num_got"])
        print("  row", r, "ref", np.round(a, 6).tolist()); print("         got", np.round(g, 6).tolist()); print("         got-ref", np.round(g - a, 6).tolist())
