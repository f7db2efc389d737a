"""The per-round performance matrix: every shape class the schedule layer distinguishes, DEVICE-timed,
min of five, written beside the previous round's column.

    python tools/perf_matrix.py [--out profiles/r06/perf_matrix.json] [--rows name,name,...] [--reps 5] [--quick]

Not a test: nothing in `pytest -m gpu` depends on a clock.  Every schedule change re-runs this and commits the file.

How a row is timed.  NMF rows: `fluhip_corpus_last_loop_ms` -- two HIP events on the context stream, the first behind the
host-side initialisation of `fluhip_corpus_nmf`, the second behind the last launch of its iteration loop -- divided by the
iterations of the call; `reps` calls, the MINIMUM is the figure (the mean and the spread are kept).  No host scheduling,
no random draws, no upload is inside it.  Feature / STFT rows: HIP events recorded by this process on the context
stream (`fluhip_ctx_stream`) around the call, inputs and outputs resident.  The sustained shader clock of the update
launches (s_memtime / s_memrealtime stamps) is kept per row where the schedule carries the stamps, so that two boxes
can be compared at equal clock.

`prev` holds what round 5 left on file for the same shape (profiles/r05, BENCH_r05.json, GPUTEST_r05.json), `None` where
round 5 never measured it -- that gap is how the ragged regression went unseen.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flucoma-core_amd"))
import fluhip  # noqa: E402
import synth  # noqa: E402

SR = 44100


class DevTimer:
    """HIP events on the library's own stream, through the runtime the library already loaded"""

    def __init__(self, ctx):
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self.hip.hipEventSynchronize.argtypes = [ctypes.c_void_p]
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        ctx.lib.fluhip_ctx_stream.restype = ctypes.c_void_p
        ctx.lib.fluhip_ctx_stream.argtypes = [ctypes.c_void_p]
        self.stream = ctypes.c_void_p(ctx.lib.fluhip_ctx_stream(ctx.h))
        self.e0, self.e1 = ctypes.c_void_p(), ctypes.c_void_p()
        assert self.hip.hipEventCreate(ctypes.byref(self.e0)) == 0 and self.hip.hipEventCreate(ctypes.byref(self.e1)) == 0

    def time(self, fn):
        assert self.hip.hipEventRecord(self.e0, self.stream) == 0
        fn()
        assert self.hip.hipEventRecord(self.e1, self.stream) == 0
        assert self.hip.hipEventSynchronize(self.e1) == 0
        ms = ctypes.c_float(0)
        assert self.hip.hipEventElapsedTime(ctypes.byref(ms), self.e0, self.e1) == 0
        return float(ms.value)


def stats(xs):
    xs = [float(x) for x in xs]
    return {"min": min(xs), "mean": sum(xs) / len(xs), "max": max(xs), "n": len(xs)}


def nmf_loop_us(ctx, cor, iters, reps):
    cor.nmf(max(4, iters // 4), seed=42); ctx.synchronize()          # warm: clocks up, code objects resident, blocks cached
    cor.update_clocks(reset=True)
    per = []
    for _ in range(reps):
        cor.nmf(iters, seed=42)
        per.append(cor.last_loop_ms() * 1e3 / iters)
    clk = cor.update_clocks(reset=True)
    mhz = [v["sustained_mhz"] for v in clk.values() if v["sustained_mhz"]]
    return stats(per), (sum(mhz) / len(mhz) if mhz else None)


def corpus_row(ctx, B, seconds, K, iters, reps, win=2048, fft=2048, hop=512, tile=None):
    n = int(seconds * SR)
    if tile:                                                          # long inputs: a 10 s clip tiled (synthesis is host time)
        base = [synth.synth_audio(tile, 1000 + b) for b in range(min(B, 8))]
        x = np.stack([np.tile(base[b % len(base)], n // tile + 1)[:n] for b in range(B)])
    else:
        base = [synth.synth_audio(n, 1000 + b) for b in range(min(B, 8))]
        x = np.stack([base[b % len(base)] for b in range(B)])
    cor = fluhip.Corpus(ctx, B, n, win, fft, hop, K)
    cor.set_audio(x); cor.stft(); ctx.synchronize()
    del x
    us, mhz = nmf_loop_us(ctx, cor, iters, reps)
    T, F = cor.T, cor.F
    flop = 8.0 * F * T * K * B
    nbytes = (2.0 * F * T + 4.0 * (F * K + K * T)) * 8.0 * B
    row = {"unit": "us per iteration (all buffers advance one iteration), device-timed", "us": us, "iterations_per_call": iters,
           "shape": {"buffers": B, "frames": T, "bins": F, "rank": K}, "plan": cor.plan(), "sustained_mhz": mhz,
           "tflops": flop / (us["min"] * 1e-6) / 1e12, "frac_fp64_matrix": flop / (us["min"] * 1e-6) / 1e12 / 78.6,
           "gbs_algorithmic": nbytes / (us["min"] * 1e-6) / 1e9, "frac_hbm": nbytes / (us["min"] * 1e-6) / 1e9 / 8000.0}
    cor.close()
    return row


def ragged_lens(B, nd, lo, hi):
    rs = np.random.RandomState(7)
    distinct = sorted(int(x) for x in rs.randint(int(lo * SR), int(hi * SR), nd))
    lens = [distinct[i % nd] for i in range(B)]
    rs.shuffle(lens)
    return lens


def ragged_rows(ctx, B, nd, lo, hi, K, iters, reps):
    """the ragged corpus and its equal-length twin (same number of buffers, same mean length): the review's ratio"""
    win, fft, hop = 2048, 2048, 512
    lens = ragged_lens(B, nd, lo, hi)
    base = [synth.synth_audio(int(hi * SR), 5000 + i) for i in range(8)]
    c = fluhip.RaggedCorpus(ctx, lens, win, fft, hop, K)
    c.set_audio([base[i % 8][:n] for i, n in enumerate(lens)]); c.stft(); ctx.synchronize()
    us_r, mhz_r = nmf_loop_us(ctx, c, iters, reps)
    plan_r = c.plan()
    frames = int(sum(c.Ts))
    c.close()
    n_eq = int(sum(lens) / len(lens))
    u = fluhip.Corpus(ctx, B, n_eq, win, fft, hop, K)
    u.set_audio(np.stack([base[i % 8][:n_eq] for i in range(B)])); u.stft(); ctx.synchronize()
    us_e, mhz_e = nmf_loop_us(ctx, u, iters, reps)
    plan_e = u.plan()
    u.close()
    unit = "us per iteration (all buffers advance one iteration), device-timed"
    return ({"unit": unit, "us": us_r, "iterations_per_call": iters, "plan": plan_r, "sustained_mhz": mhz_r,
             "shape": {"buffers": B, "distinct_lengths": nd, "seconds": [lo, hi], "frames_total": frames, "rank": K},
             "ratio_to_equal_length_twin": us_r["min"] / us_e["min"]},
            {"unit": unit, "us": us_e, "iterations_per_call": iters, "plan": plan_e, "sustained_mhz": mhz_e,
             "shape": {"buffers": B, "samples": n_eq, "rank": K}})


def c5_row(ctx, tm, reps):
    """BASELINE config 5 with audio and coefficients resident (torch holds the device arrays)"""
    import torch
    count, n, win, fft, hop, nb, nc = 8192, 88200, 1024, 1024, 512, 40, 13
    base = np.stack([synth.synth_audio(n, 1000 + b) for b in range(64)])
    a_dev = torch.from_numpy(np.tile(base, (count // 64, 1))).cuda()
    T = n // hop + 1
    o_dev = torch.empty((count, nc, T), dtype=torch.float32, device="cuda")
    Tr = ctypes.c_int64(0)

    def run():
        rc = ctx.lib.fluhip_bufmfcc_f32(ctx.h, ctypes.cast(a_dev.data_ptr(), ctypes.POINTER(ctypes.c_float)), count, n,
                                        win, fft, hop, nb, nc, 0, 20.0, 20000.0, 44100.0,
                                        ctypes.cast(o_dev.data_ptr(), ctypes.POINTER(ctypes.c_float)), ctypes.byref(Tr))
        assert rc == 0
    torch.cuda.synchronize()
    run(); ctx.synchronize()
    ms = stats([tm.time(run) for _ in range(reps)])
    assert Tr.value == T
    return {"unit": "ms per call (8192 x 2 s -> 13 MFCCs per frame), device-timed", "ms": ms,
            "shape": {"slices": count, "frames_per_slice": T}, "frames_per_s": count * T / (ms["min"] * 1e-3)}


def stft_row(ctx, tm, reps):
    """the STFT phase of the bench shard (both magnitude layouts: what the NMF corpus path runs)"""
    B, n, win, fft, hop, K = 128, 441000, 2048, 2048, 512, 32
    base = [synth.synth_audio(n, 1000 + b) for b in range(8)]
    cor = fluhip.Corpus(ctx, B, n, win, fft, hop, K)
    cor.set_audio(np.stack([base[b % 8] for b in range(B)])); cor.stft(); ctx.synchronize()
    ms = stats([tm.time(cor.stft) for _ in range(reps)])
    T, F = cor.T, cor.F
    nbytes = (hop * 4.0 + F * 8.0) * T * B
    cor.stft_mag_only(); ctx.synchronize()
    ms1 = stats([tm.time(cor.stft_mag_only) for _ in range(reps)])
    cor.close()
    return {"unit": "ms per STFT phase of the bench shard, device-timed", "ms": ms, "frames_per_s": B * T / (ms["min"] * 1e-3),
            "frac_hbm_algorithmic": nbytes / (ms["min"] * 1e-3) / 8e12,
            "single_layout": {"ms": ms1, "frames_per_s": B * T / (ms1["min"] * 1e-3), "frac_hbm": nbytes / (ms1["min"] * 1e-3) / 8e12}}


def client_row(driver, reps, tmp):
    """the 8-channel x 10 s rank-32 BufNMF job through the C++17 host client (wall time of process(): host timed by necessity
    -- it is a host-side job: gather, upload, 200 iterations, write-back -- min of `reps`)"""
    import subprocess
    frames, chans = 441000, 8
    audio = np.stack([synth.synth_audio(frames, 800 + c) for c in range(chans)], axis=1).astype(np.float32)
    inp = os.path.join(tmp, "in.f32")
    audio.tofile(inp)
    out = {}
    for tag, env in (("batched", {}), ("sequential", {"FLUHIP_CLIENT_SEQUENTIAL": "1"})):
        e = dict(os.environ); e.update(env); e["CLIENT_REPEAT"] = str(reps + 1); e["CLIENT_REPEAT_PRINT"] = "1"
        p = subprocess.run([driver, "run", inp, str(frames), str(chans), "2048", "512", "2048", "32", "200", "42", "0", "0", "0", "0",
                            "-1", "0", "-1", os.path.join(tmp, tag)], capture_output=True, text=True, timeout=600, env=e)
        if p.returncode != 0:
            return {"error": p.stderr[-500:]}
        el = [float(line.split(":")[1].split()[0]) for line in p.stderr.splitlines() if line.startswith("repeat ")]
        if len(el) < 2:
            return {"error": "no repeat lines: " + p.stderr[-300:]}
        out[tag] = stats(el[1:])
    return {"unit": "ms per 8-channel job (host wall time of the client's process(), first call dropped)", **out,
            "batched_over_sequential": out["batched"]["min"] / out["sequential"]["min"]}


# round 5's figures for the same shapes, with where they are on file (None: round 5 never measured the shape)
PREV = {
    "bench_shard_128x10s_k32": {"us": 566.2, "src": "BENCH_r05.json: 113.23 ms per 200-iteration step (host-timed, STFT + write-back inside: ~0.6 ms)"},
    "c2_60s_k16": {"us": 38.72, "src": "BENCH_r05.json configs.c2"},
    "c3_2x10min_k128_fft4096": {"us": 1845.5, "src": "BENCH_r05.json configs.c3"},
    "c4x1_10s_k32": {"us": None, "src": "not measured in r05"},
    "c1_shape_k3": {"us": 20.6, "src": "DESIGN r05 K7"},
    "c5_mfcc_8192x2s": {"ms": 2.917, "src": "BENCH_r05.json configs.c5"},
    "stft_bench_shard": {"ms": 0.5641, "src": "BENCH_r05.json roofline_stft (kernel launch, rocprof)"},
    "ragged_64x40": {"us": 420.0, "src": "GPUTEST_r05.json (driver's box, one host-timed sample); 341 us in profiles/r03/ragged_64x40_v2.json"},
    "ragged_64x40_equal_twin": {"us": 290.0, "src": "GPUTEST_r05.json (same sample)"},
    "ragged_256x100": {"us": None, "src": "not measured in r05"},
    "ragged_256x100_equal_twin": {"us": None, "src": "not measured in r05"},
    "corpus_128x10s_k20": {"us": 480.0, "src": "profiles/r05/offsize_ranks.txt: 24.0 ms per 50 iterations"},
    "corpus_128x10s_k40": {"us": 782.0, "src": "profiles/r05/offsize_ranks.txt: 39.1 ms per 50"},
    "corpus_128x10s_k96": {"us": 1882.0, "src": "profiles/r05/offsize_ranks.txt: 94.1 ms per 50"},
    "corpus_128x10s_k128": {"us": 2204.0, "src": "DESIGN r05: 110.2 ms per 50"},
    "corpus_8x10s_k32": {"us": 73.0, "src": "tests/test_client.py docstring (r03: 200 iterations at 73 us)"},
    "long_60s_k64": {"us": None, "src": "not measured in r05"},
    "single_10s_k128": {"us": 104.5, "src": "DESIGN r05: 20.9 ms per 200 iterations"},
    "client_8ch_10s_k32": {"ms": 22.3, "src": "tests/test_client.py docstring (r03)"},
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--rows", default=None)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--quick", action="store_true", help="a tenth of the iterations per call (smoke run of the tool itself)")
    a = ap.parse_args()
    q = 10 if a.quick else 1
    ctx = fluhip.Context(0)
    tm = DevTimer(ctx)
    name, arch, cus = ctx.device_info()
    rows = {}
    R = a.reps

    def want(n):
        return a.rows is None or n in a.rows.split(",")

    def put(n, fn):
        if not want(n):
            return
        t0 = time.perf_counter()
        try:
            r = fn()
        except Exception as ex:      # one failing row must not cost the others
            r = {"error": f"{type(ex).__name__}: {ex}"}
        r["wall_s"] = time.perf_counter() - t0
        r["prev"] = PREV.get(n)
        rows[n] = r
        print(n, json.dumps({k: v for k, v in r.items() if k in ("us", "ms", "error", "ratio_to_equal_length_twin", "sustained_mhz", "single_layout", "batched", "sequential")}),
              file=sys.stderr, flush=True)

    put("bench_shard_128x10s_k32", lambda: corpus_row(ctx, 128, 10, 32, 200 // q, R))
    put("c2_60s_k16", lambda: corpus_row(ctx, 1, 60, 16, 200 // q, R, tile=441000))
    put("c1_shape_k3", lambda: corpus_row(ctx, 1, 453932 / SR, 3, 200 // q, R, win=1024, fft=1024, hop=512))
    put("c4x1_10s_k32", lambda: corpus_row(ctx, 1, 10, 32, 200 // q, R))
    put("corpus_8x10s_k32", lambda: corpus_row(ctx, 8, 10, 32, 200 // q, R))
    for K in (20, 40, 96, 128):
        put(f"corpus_128x10s_k{K}", lambda K=K: corpus_row(ctx, 128, 10, K, 50 // min(q, 5), R))
    put("long_60s_k64", lambda: corpus_row(ctx, 1, 60, 64, 100 // q, R, tile=441000))
    put("single_10s_k128", lambda: corpus_row(ctx, 1, 10, 128, 200 // q, R))
    if want("ragged_64x40") or want("ragged_64x40_equal_twin"):
        t0 = time.perf_counter()
        try:
            rr, re_ = ragged_rows(ctx, 64, 40, 4.0, 16.0, 32, 100 // q, R)
        except Exception as ex:
            rr = re_ = {"error": f"{type(ex).__name__}: {ex}"}
        for n, r in (("ragged_64x40", rr), ("ragged_64x40_equal_twin", re_)):
            r = dict(r); r["prev"] = PREV.get(n); r["wall_s"] = time.perf_counter() - t0; rows[n] = r
        print("ragged_64x40", json.dumps({k: rr.get(k) for k in ("us", "ratio_to_equal_length_twin", "error")}), file=sys.stderr, flush=True)
    if want("ragged_256x100") or want("ragged_256x100_equal_twin"):
        t0 = time.perf_counter()
        try:
            rr, re_ = ragged_rows(ctx, 256, 100, 2.0, 12.0, 32, 60 // min(q, 6), R)
        except Exception as ex:
            rr = re_ = {"error": f"{type(ex).__name__}: {ex}"}
        for n, r in (("ragged_256x100", rr), ("ragged_256x100_equal_twin", re_)):
            r = dict(r); r["prev"] = PREV.get(n); r["wall_s"] = time.perf_counter() - t0; rows[n] = r
        print("ragged_256x100", json.dumps({k: rr.get(k) for k in ("us", "ratio_to_equal_length_twin", "error")}), file=sys.stderr, flush=True)
    put("stft_bench_shard", lambda: stft_row(ctx, tm, R))
    put("c5_mfcc_8192x2s", lambda: c5_row(ctx, tm, R))
    put("c3_2x10min_k128_fft4096", lambda: corpus_row(ctx, 2, 600, 128, 40 // min(q, 4), max(3, R - 2), win=4096, fft=4096, hop=1024, tile=441000))
    if want("client_8ch_10s_k32"):
        import importlib.util
        import tempfile
        spec = importlib.util.spec_from_file_location("fluhip_build", os.path.join(ROOT, "flucoma-core_amd", "build.py"))
        mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
        with tempfile.TemporaryDirectory() as tmp:
            put("client_8ch_10s_k32", lambda: client_row(mod.build_host_tests(), R, tmp))
    out = {"tool": "tools/perf_matrix.py", "device": {"name": name, "arch": arch, "compute_units": cus},
           "timing": "device (HIP events on the library's stream), min of reps; client row: host wall time of process()",
           "reps": R, "quick": a.quick, "rows": rows}
    txt = json.dumps(out, indent=1)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            f.write(txt + "\n")
    print(txt)
    ctx.close()


if __name__ == "__main__":
    main()
