"""Reproducer sweep for the ragged rank-40 mismatch (round 6, session 2): W / H relative errors against the oracle per buffer.
    python tools/probes_ragged_k40.py K uw uh iters [equal T]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, 'flucoma-core_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'));
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
if os.environ.get('FLUHIP_PY_DIR'): sys.path.insert(0, os.environ['FLUHIP_PY_DIR'])
import numpy as np
import test_gpu_random_shapes as t, fluhip
import oracle_c, oracle_np as onp
from helpers import rel_err
case = [c for c in t._ragged_cases() if c[1] == 20 and c[6] == 40][0]
_, B, _, win, fft, hop, K, iters, uw, uh, lens = case
lens = list(lens)
K = int(sys.argv[1]); uw = bool(int(sys.argv[2])); uh = bool(int(sys.argv[3])); iters = int(sys.argv[4])
if len(sys.argv) > 5 and sys.argv[5] == "equal":
    lens = [int(sys.argv[6]) * hop - 5] * B
orc = oracle_c.Oracle()
ctx = fluhip.Context(0)
src = [onp.synth_audio(max(lens), 8100 + b) for b in range(3)]
audios = [src[b % 3][:n] for b, n in enumerate(lens)]
if len(set(lens)) == 1:
    c = fluhip.Corpus(ctx, B, lens[0], win, fft, hop, K); c.set_audio(np.stack(audios))
else:
    c = fluhip.RaggedCorpus(ctx, lens, win, fft, hop, K); c.set_audio(audios)
c.stft(); c.nmf(iters, seed=42, updateW=uw, updateH=uh)
mag, W1, H1 = c.read_f64(); plan = c.plan(); c.close()
out = []
for b in range(B):
    T = (lens[b] + hop) // hop
    _, rmag = orc.stft_f32(audios[b], win, fft, hop)
    rW, rH, _, _ = orc.nmf_process(rmag, K, iters, uw, uh, 42)
    out.append("%d:T%d:%.0e/%.0e" % (b, T, rel_err(W1[b], rW), rel_err(H1[b, :T], rH)))
print(sys.argv[1:], os.environ.get("FLUHIP_LIB", ""), {k: plan[k] for k in ("split_w", "split_h", "strips_w", "compute_rank", "padded_rank")}, " ".join(out))
if len(sys.argv) > 7 and sys.argv[7] == "pattern":
    b = 0; T = (lens[b] + hop) // hop
    _, rmag = orc.stft_f32(audios[b], win, fft, hop)
    rW, rH, _, _ = orc.nmf_process(rmag, K, iters, uw, uh, 42)
    E = np.abs(W1[b] - rW) / np.abs(rW).max()
    print("W shape", W1[b].shape, "err by component:", ["%.0e" % e for e in E.max(axis=1 if E.shape[0] == K else 0)])
    eb = E.max(axis=0 if E.shape[0] == K else 1)
    print("err by bin (first 70):", ["%.0e" % e for e in eb[:70]])
    print("bins with err > 1e-9:", int((eb > 1e-9).sum()), "of", eb.size, "first bad", np.nonzero(eb > 1e-9)[0][:40])
    Wt = W1[b] if W1[b].shape[0] == K else W1[b].T
    rWt = rW if rW.shape[0] == K else rW.T
    R = Wt / rWt
    np.set_printoptions(precision=5, linewidth=220)
    print("ratio rows k=0,1,39 bins 0..7:\n", R[[0, 1, 39], :8], "\n bins 512..519:\n", R[[0, 1, 39], 512:520], "\n bins 1017..1024:\n", R[[0, 1, 39], 1017:1025])
    print("per-component ratio spread (max/min over bins):", (R.max(axis=1) / R.min(axis=1))[:10])
    print("per-bin ratio spread over components:", (R.max(axis=0) / R.min(axis=0))[:10])
