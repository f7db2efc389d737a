"""Reproducer / bisecting harness for the in-place pipeline form of nmf_update5_kernel (MODE 2; DESIGN section 3
"Wide ranks").  One process = one library + one environment (the A/B switches are read once per process):

    FLUHIP_LIB=<lib> [FLUHIP_K5_MODE=.. FLUHIP_K5_MODE_ANY=1] python tools/mode2_bisect.py --tag T --rank 32 \
        [--samples 70000] [--buffers 128] [--repeats 3] [--save ref.npz | --ref ref.npz]

Cases per repeat: one W update alone, one H update alone, four full iterations.  The corpus holds `--distinct` different
inputs replicated over the buffers with ONE seed, so replicas must come out bit-identical: every buffer is compared with
the first replica of its input (exact), and with the reference file when one is given (relative, printed with the
positions that differ).  Positions are reported as (row, component) of W (bins x rank) and H (frames x rank)."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flucoma-core_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import fluhip  # noqa: E402
import oracle_np  # noqa: E402


def positions(a, b, tol):
    """(row, component) of entries of a that differ from b by more than tol * max|b|"""
    d = np.abs(a - b) > tol * max(np.abs(b).max(), 1e-300)
    return [(int(r), int(c)) for r, c in zip(*np.nonzero(d))]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--rank", type=int, default=32)
    ap.add_argument("--samples", type=int, default=70000)
    ap.add_argument("--buffers", type=int, default=128)
    ap.add_argument("--distinct", type=int, default=4)
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--fft", type=int, default=2048)
    ap.add_argument("--cases", default="w_only,h_only,full")
    ap.add_argument("--save")
    ap.add_argument("--ref")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "mode2"))
    a = ap.parse_args()
    ctx = fluhip.Context(0)
    B, D = a.buffers, a.distinct
    audio = np.stack([oracle_np.synth_audio(a.samples, 1100 + (b % D)) for b in range(B)])
    ref = np.load(a.ref) if a.ref else None
    rec = {"tag": a.tag, "lib": os.environ.get("FLUHIP_LIB", "production"), "rank": a.rank, "samples": a.samples,
           "buffers": B, "env": {k: v for k, v in os.environ.items() if k.startswith("FLUHIP_K5")}, "cases": []}
    saved = {}
    for rep in range(a.repeats):
        for case, (uw, uh, it) in {"w_only": (True, False, 1), "h_only": (False, True, 1), "full": (True, True, a.iters)}.items():
            if case not in a.cases.split(","):
                continue
            c = fluhip.Corpus(ctx, B, a.samples, a.fft, a.fft, a.fft // 4, a.rank)
            c.set_audio(audio); c.stft(); c.nmf(it, seed=42, updateW=uw, updateH=uh)
            _, W, H = c.read_f64(mag=False)
            W = np.ascontiguousarray(W.transpose(0, 2, 1))   # (buffer, bin, component), like H (buffer, frame, component)
            plan = c.plan()
            c.close()
            entry = {"case": case, "repeat": rep, "plan": {k: plan[k] for k in plan if not isinstance(plan[k], (list, dict))}}
            for name, M in (("W", W), ("H", H)):
                bad = {}
                for b in range(B):
                    if not np.array_equal(M[b], M[b % D]):
                        bad[b] = positions(M[b], M[b % D], 0.0)
                entry[name + "_replica_mismatch_buffers"] = len(bad)
                entry[name + "_replica_mismatch"] = {str(b): p[:24] for b, p in list(bad.items())[:12]}
                if ref is not None:
                    R = ref[f"{case}_{name}"]
                    worst, where = 0.0, {}
                    for b in range(B):
                        e = float(np.abs(M[b] - R[b % D]).max() / max(np.abs(R[b % D]).max(), 1e-300))
                        worst = max(worst, e)
                        if e > 1e-9 and len(where) < 12:
                            where[str(b)] = {"err": e, "at": positions(M[b], R[b % D], 1e-9)[:24]}
                    entry[name + "_vs_ref_worst"] = worst
                    entry[name + "_vs_ref_bad"] = where
                if rep == 0:
                    saved[f"{case}_{name}"] = M[:D].copy()
            rec["cases"].append(entry)
            print(json.dumps(entry)[:1500], flush=True)
    os.makedirs(a.out, exist_ok=True)
    with open(os.path.join(a.out, a.tag + ".json"), "w") as f:
        json.dump(rec, f, indent=1)
    if a.save:
        np.savez(a.save, **saved)
    nbad = sum(e["W_replica_mismatch_buffers"] + e["H_replica_mismatch_buffers"] for e in rec["cases"])
    nref = sum(1 for e in rec["cases"] for n in ("W", "H") if e.get(n + "_vs_ref_worst", 0.0) > 1e-9)
    print(f"SUMMARY {a.tag}: replica-mismatching buffers {nbad}, cases off the reference {nref}")


if __name__ == "__main__":
    main()
