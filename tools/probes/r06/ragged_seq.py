"""All ragged cases of tests/test_gpu_random_shapes.py in ONE process, in order (or only those named by index): max W / H error per case.
    [FLUHIP_PY_DIR=.. FLUHIP_LIB=..] python tools/probes_ragged_seq.py [i j k ...]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
for p in ('flucoma-core_amd', 'tests', 'oracle'): sys.path.insert(0, os.path.join(ROOT, p))
if os.environ.get('FLUHIP_PY_DIR'): sys.path.insert(0, os.environ['FLUHIP_PY_DIR'])
import numpy as np
import test_gpu_random_shapes as t, fluhip
import oracle_c, oracle_np as onp
from helpers import rel_err
orc = oracle_c.Oracle(); ctx = fluhip.Context(0)
want = [int(a) for a in sys.argv[1:]]
for case in t._ragged_cases():
    i, B, _, win, fft, hop, K, iters, uw, uh, lens = case
    if want and i not in want: continue
    lens = list(lens)
    src = [onp.synth_audio(max(lens), 8100 + b) for b in range(3)]
    audios = [src[b % 3][:n] for b, n in enumerate(lens)]
    c = fluhip.RaggedCorpus(ctx, lens, win, fft, hop, K)
    c.set_audio(audios); c.stft(); c.nmf(iters, seed=42, updateW=uw, updateH=uh)
    mag, W1, H1 = c.read_f64(); plan = c.plan(); c.close()
    ew = eh = 0.0
    for b in sorted({0, B - 1, int(np.argmin(lens)), int(np.argmax(lens))}):
        T = (lens[b] + hop) // hop
        _, rmag = orc.stft_f32(audios[b], win, fft, hop)
        rW, rH, _, _ = orc.nmf_process(rmag, K, iters, uw, uh, 42)
        ew = max(ew, rel_err(W1[b], rW)); eh = max(eh, rel_err(H1[b, :T], rH))
    print("case %d B%d K%d fft%d hop%d it%d uw%d uh%d Tmax%d: W %.0e H %.0e  strips_w %s kc %s" % (i, B, K, fft, hop, iters, uw, uh, (max(lens) + hop) // hop, ew, eh, plan.get('strips_w'), plan.get('compute_rank')), flush=True)
