"""One JSON line per BASELINE config (c1 .. c5), through the C ABI, with the roofline computed as in bench.py.

    python tools/bench_configs.py [c1] [c2] [c3] [c4x1] [c5] [resynth] [--no-cpu] [--prof]

(bench.py imports this module to run c2 / c3 / c5 behind its headline and put their driver-timed figures into the same
JSON line: `configs`.  It lives under tools/, not in the package, because its cpu_baseline legs call the oracle.)

The configs other than c4 are parity-test shapes, not bench.py lines; this tool is what DESIGN section 6 quotes for
them and what the rocprofv3 summaries under profiles/ were taken on (tools/profile_configs.sh).  Work per unit is
SURVEY 8(d)'s: one NMF iteration = 8 F T K flop and (2 F T + 4 (F K + K T)) 8 bytes; one STFT frame = hop 4 + F 8
bytes; one MFCC frame (c5) = hop 4 bytes in + nCoefs 4 bytes out.  Times are wall-clock around a drained stream.
The per-iteration cost is (one call of `timed` >= 100 iterations, after a warm call) minus (a call of 0 iterations:
the host-side factor initialisation, a fixed cost per call), divided by `timed` -- launch gaps between the kernels of an
iteration are inside it, and the tool REFUSES a figure below the sum of the kernels' own mean durations over the same
number of iterations (HIP events): round 2's slope between a 5- and a 20-iteration call did fall below it.
`cpu_baseline` is the oracle (a restatement, not the Eigen binary) on one host core over a bounded sample.
"""
import json
import os
import sys
import time

import numpy as np

try:  # torch first when it is there: the two then share one HIP runtime (config 5's device-resident variant)
    import torch  # noqa: F401
except ImportError:
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flucoma-core_amd"))
import fluhip  # noqa: E402
import synth  # noqa: E402

PEAK_HBM_GBS = 8000.0
PEAK_FP64_TFLOPS = 78.6

CONFIGS = {
    # name: samples, channels/buffers, win, fft, hop, rank, iterations of the config, iterations timed here
    "c1": dict(n=453932, B=1, win=1024, fft=1024, hop=512, K=3, iters=50, timed=50,
               what="BASELINE config 1 shape (10.3 s mono, the bundled loop's length), rank 3, 50 iterations"),
    "c2": dict(n=2646000, B=1, win=2048, fft=2048, hop=512, K=16, iters=200, timed=200,
               what="BASELINE config 2: 60 s mono, fft 2048 / hop 512, rank 16, 200 iterations"),
    "c3": dict(n=26460000, B=2, win=4096, fft=4096, hop=1024, K=128, iters=500, timed=120,
               what="BASELINE config 3: 10 min stereo (2 channels resident as one corpus), fft 4096 / hop 1024, "
                    "rank 128; 120 of the 500 iterations timed in one call"),
    "c4x1": dict(n=441000, B=1, win=2048, fft=2048, hop=512, K=32, iters=200, timed=200,
                 what="one buffer of BASELINE config 4 on its own: 10 s mono, rank 32, 200 iterations"),
}


def device_line(ctx):
    name, arch, cus = ctx.device_info()
    return {"name": name, "arch": arch, "compute_units": cus}


def cpu_nmf(mag, K, seed, budget_s=12.0):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_c
    o = oracle_c.get("native")
    t0 = time.perf_counter()
    o.nmf_process(mag, K, 1, True, True, seed, faithful=True)
    one = time.perf_counter() - t0
    it = int(max(1, min(20, budget_s / max(one, 1e-9))))
    t0 = time.perf_counter()
    o.nmf_process(mag, K, it, True, True, seed, faithful=True)
    dt = time.perf_counter() - t0
    return {"value": it / dt, "unit": "iterations/s", "cores": 1, "kind": "port",
            "sample": f"{it} iterations of one channel, oracle faithful mode (7 GEMMs per iteration), gcc -O3 -march=native"}


def run_nmf_config(ctx, name, with_cpu):
    c = CONFIGS[name]
    B, n, K = c["B"], c["n"], c["K"]
    # a 10 s synthetic clip per channel, tiled to the requested length
    chans = []
    for b in range(B):
        base = synth.synth_audio(min(n, 441000), 1000 + b)
        chans.append(np.tile(base, n // len(base) + 1)[:n])
    x = np.stack(chans)
    cor = fluhip.Corpus(ctx, B, n, c["win"], c["fft"], c["hop"], K)
    cor.set_audio(x)
    cor.stft(); ctx.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        cor.stft()
    ctx.synchronize()
    t_stft = (time.perf_counter() - t0) / reps
    ctx.prof_enable(True); ctx.prof_reset()
    cor.stft(); ctx.synchronize()
    _, stft_kernel_ms = ctx.prof_read(0)
    ctx.prof_enable(False)
    T, F = cor.T, cor.F
    n2 = max(100, c["timed"])
    cor.nmf(20, seed=42); ctx.synchronize()            # warm call: clocks up, allocations cached
    t0s = []
    for _ in range(3):                                 # the fixed cost of a call: initialisation only
        t0 = time.perf_counter(); cor.nmf(0, seed=42); ctx.synchronize(); t0s.append(time.perf_counter() - t0)
    t_fixed = min(t0s)
    cor.update_clocks(reset=True)
    t0 = time.perf_counter(); cor.nmf(n2, seed=42); ctx.synchronize(); t2 = time.perf_counter() - t0
    clocks = cor.update_clocks(reset=True)
    per_it = (t2 - t_fixed) / n2                        # all B channels advance one iteration
    fixed_ms = t_fixed * 1e3
    # kernel-only view of the same loop (HIP events on the context's stream), over the same number of iterations
    ctx.prof_enable(True); ctx.prof_reset()
    cor.nmf(n2, seed=42); ctx.synchronize()
    n_upd, ms_upd = ctx.prof_read(1)
    n_mid, ms_mid = ctx.prof_read(3)
    ctx.prof_enable(False)
    n1 = n2
    kernels_us = (ms_upd + ms_mid) / n2 * 1e3
    # (kernels of a few microseconds: the two event records around each one are a visible part of its event-timed
    #  duration, so the floor is only enforced where launches are long against that -- config 3's millisecond launches)
    if ms_upd / max(n_upd, 1) > 0.1:
        assert per_it * 1e6 >= 0.97 * kernels_us, (
            f"{name}: {per_it * 1e6:.1f} us per iteration is below the kernels' own {kernels_us:.1f} us -- not a measurement")
    flop_it = 8.0 * F * T * K * B
    bytes_it = (2.0 * F * T + 4.0 * (F * K + K * T)) * 8.0 * B
    tf = flop_it / per_it / 1e12
    gbs = bytes_it / per_it / 1e9
    ai = flop_it / bytes_it
    bound = "mfma" if ai > PEAK_FP64_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9) else "hbm"
    roof = ({"bound": "mfma", "achieved": tf, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_FP64_TFLOPS}
            if bound == "mfma" else
            {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS})
    roof.update({"per": "NMF iteration (all launches of it, gaps included)", "flop": flop_it, "bytes": bytes_it,
                 "other_view": {"TFLOP/s": tf, "GB/s": gbs}, "traffic": None,
                 # the update launches alone (HIP events over the same iterations; what a rocprofv3 kernel-stats CSV
                 # of this command reproduces: flop per iteration / (launches per iteration x mean launch duration))
                 "update_kernels_only": {"TFLOP/s": flop_it / (ms_upd / n2 * 1e-3) / 1e12,
                                         "frac_of_fp64_matrix_peak": flop_it / (ms_upd / n2 * 1e-3) / 1e12 / PEAK_FP64_TFLOPS,
                                         "GB/s": bytes_it / (ms_upd / n2 * 1e-3) / 1e9}})
    stft_bytes = (c["hop"] * 4.0 + F * 8.0) * T * B
    out = {
        "config": name, "workload": c["what"],
        "metric": "NMF iterations/s (channel-iterations; all channels advance together)",
        "value": B / per_it, "unit": "iterations/s", "higher_is_better": True, "dtype": "f64", "data": "synthetic",
        "us_per_iteration": per_it * 1e6, "fixed_ms_per_call": fixed_ms,
        "nmf_job_ms_est": (fixed_ms + per_it * c["iters"] * 1e3),
        "kernel_ms_per_iteration": {"updates": ms_upd / n1, "between": ms_mid / n1,
                                    "update_launches_per_iteration": n_upd / n1,
                                    "update_launch_mean_ms": ms_upd / max(n_upd, 1)},
        "update_clocks": clocks,
        "stft_frames_per_s": T * B / t_stft, "stft_ms": t_stft * 1e3, "stft_kernel_ms": stft_kernel_ms,
        "shape": {"channels": B, "samples": n, "frames": T, "bins": F, "rank": K, "iterations": c["iters"],
                  "iterations_timed": n2},
        "roofline": roof,
        "roofline_stft": {"bound": "hbm", "achieved": stft_bytes / t_stft / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                          "frac": stft_bytes / t_stft / 1e9 / PEAK_HBM_GBS, "per": "STFT phase (every launch of it)",
                          "bytes": stft_bytes},
        "schedule": cor.plan(), "device": dict(device_line(ctx), corpus_device_bytes=cor.device_bytes()),
    }
    if with_cpu:
        mag = cor.read_f64(mag=True, factors=False)[0][0] if B * T * F * 8 < (1 << 30) else None
        if mag is None:   # c3: one channel's magnitudes come back alone
            one = fluhip.Corpus(ctx, 1, n, c["win"], c["fft"], c["hop"], K)
            one.set_audio(x[:1]); one.stft()
            mag = one.read_f64(mag=True, factors=False)[0][0]
            one.close()
        out["cpu_baseline"] = cpu_nmf(np.ascontiguousarray(mag), K, 42)
        out["cpu_baseline"]["gpu_speedup_per_channel_iteration"] = (1.0 / per_it) / out["cpu_baseline"]["value"]
    cor.close()
    return out


def run_c5(ctx, with_cpu):
    """BASELINE config 5: STFT -> MelBands(40) -> MFCC(13) over 8192 x 2 s slices (clients/rt/MFCCClient.hpp defaults)."""
    import ctypes
    count, n, win, fft, hop, nb, nc = 8192, 88200, 1024, 1024, 512, 40, 13
    base = np.stack([synth.synth_audio(n, 1000 + b) for b in range(64)])
    audio = np.tile(base, (count // 64, 1))
    ctx.bufmfcc(audio[:64], win, fft, hop)
    t0 = time.perf_counter(); out = ctx.bufmfcc(audio, win, fft, hop); dt_host = time.perf_counter() - t0
    T = out.shape[2]
    frames = count * T
    res = {"config": "c5", "workload": "BASELINE config 5: STFT -> MelBands(40) -> MFCC(13) over 8192 x 2 s mono slices, "
                                       "fft 1024 / hop 512",
           "metric": "feature frames/s", "unit": "frames/s", "higher_is_better": True, "dtype": "f64", "data": "synthetic",
           "shape": {"slices": count, "samples": n, "frames_per_slice": T, "bands": nb, "coefficients": nc},
           "ms_host_buffers_in_and_out": dt_host * 1e3, "frames_per_s_host_buffers": frames / dt_host,
           "device": device_line(ctx)}
    if torch is not None:
        a_dev = torch.from_numpy(audio).cuda()
        o_dev = torch.empty(out.shape, dtype=torch.float32, device="cuda")
        Tr = ctypes.c_int64(0)

        def run():
            rc = ctx.lib.fluhip_bufmfcc_f32(ctx.h, ctypes.cast(a_dev.data_ptr(), ctypes.POINTER(ctypes.c_float)), count, n,
                                            win, fft, hop, nb, nc, 0, 20.0, 20000.0, 44100.0,
                                            ctypes.cast(o_dev.data_ptr(), ctypes.POINTER(ctypes.c_float)), ctypes.byref(Tr))
            assert rc == 0
        run(); torch.cuda.synchronize()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            run()
        torch.cuda.synchronize()
        dd = (time.perf_counter() - t0) / reps
        assert np.array_equal(o_dev.cpu().numpy(), out)
        ctx.prof_enable(True); ctx.prof_reset()
        run(); torch.cuda.synchronize()
        _, ms_stft = ctx.prof_read(0)
        _, ms_feat = ctx.prof_read(2)
        ctx.prof_enable(False)
        nbytes = (hop * 4.0 + nc * 4.0) * frames
        # Both roofs.  HBM: the samples in, the coefficients out.  FP64 vector: the arithmetic of a frame --
        # 2.5 fft log2(fft) for the half-size complex transform + split, fft for the window, 3 F for the magnitudes, the 40
        # triangular bands as two running sums over the bins (4 F), dB (~20 per band) and the DCT rows (2 bands coefficients)
        # -- against the FP64 VALU peak (= the FP64 matrix peak on this part).  The binding roof is the one that gives the
        # LONGER time; the kernel's distance from it is the fraction to read.
        F = fft // 2 + 1
        flops = frames * (2.5 * fft * np.log2(fft) + fft + 3.0 * F + 4.0 * F + 20.0 * 40 + 2.0 * 40 * nc)
        t_hbm, t_valu = nbytes / (PEAK_HBM_GBS * 1e9), flops / (PEAK_FP64_TFLOPS * 1e12)
        bound = "fp64_valu" if t_valu > t_hbm else "hbm"
        roofs = {"hbm": {"achieved": nbytes / dd / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": t_hbm / dd, "bytes": nbytes},
                 "fp64_valu": {"achieved": flops / dd / 1e12, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s", "frac": t_valu / dd, "flop": flops}}
        res.update({"value": frames / dd, "ms": dd * 1e3, "kernel_ms": {"stft": ms_stft, "features": ms_feat},
                    "roofline": {"bound": bound, **{k: v for k, v in roofs[bound].items()}, "traffic": None,
                                 "per": "whole call, audio and features resident in HBM", "roofs": roofs,
                                 "note": "the binding roof is the one whose floor is the longer time; counters put the kernel "
                                         "at 467 VALU and 134 LDS instructions per frame and wavefront (profiles/r05/cfg_v5/c5_pmc.json; 542 / 149 "
                                         "before the trims of round 5): the LDS pipe co-limits"}})
    else:
        res["value"] = frames / dt_host
    if with_cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_c
        o = oracle_c.get("native")
        t0 = time.perf_counter()
        for b in range(32):
            o.bufmfcc_channel(audio[b], win, fft, hop)
        cpu = (time.perf_counter() - t0) / 32
        res["cpu_baseline"] = {"value": T / cpu, "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": "32 of the 8192 slices, oracle BufMFCC, gcc -O3 -march=native"}
    return res


def run_resynth(ctx):
    """SURVEY 8 f1 at the bench shard's shape: every component of 128 x 10 s buffers at rank 32 resynthesised on the device
    (fluhip_corpus_resynth_dev: estimate -> ratio mask -> inverse STFT -> overlap-add, NMFClient.hpp:302-334)."""
    import ctypes
    B, n, K, win, fft, hop = 128, 441000, 32, 2048, 2048, 512
    base = np.stack([synth.synth_audio(n, 1000 + b) for b in range(4)])
    c = fluhip.Corpus(ctx, B, n, win, fft, hop, K)
    c.keep_spectrum(True)
    c.set_audio(np.tile(base, (B // 4, 1))); c.stft(); c.nmf(10, seed=42); ctx.synchronize()
    out = torch.empty((B, K, n), dtype=torch.float32, device="cuda")
    ptr = ctypes.c_void_p(out.data_ptr())
    assert ctx.lib.fluhip_corpus_resynth_dev(c.h, ptr) == 0
    ctx.synchronize()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        assert ctx.lib.fluhip_corpus_resynth_dev(c.h, ptr) == 0
    ctx.synchronize()
    dd = (time.perf_counter() - t0) / reps
    T, F = c.T, c.F
    frames = B * K * T
    err = float((out[0].sum(dim=0).cpu() - torch.from_numpy(base[0])).abs().max())
    # algorithmic bytes: the spectrum, W and H once per buffer, the float samples of every component out (SURVEY 8d style:
    # what must cross HBM once); the V-hat reciprocal the implementation stores and re-reads is extra
    nbytes = B * (T * F * 16.0 + (F + T) * K * 8.0) + B * K * n * 4.0
    flops = frames * (2.5 * fft * np.log2(fft) + 2 * fft + 6 * F)       # inverse real FFT + window / overlap-add + mask
    res = {"config": "resynth", "workload": "resynthesis of every component of the bench shard: 128 x 10 s mono, fft 2048 / hop 512, "
                                            "rank 32 (third BufNMF output, resident in HBM)",
           "metric": "component frames/s", "unit": "frames/s", "higher_is_better": True, "dtype": "f64", "data": "synthetic",
           "value": frames / dd, "ms": dd * 1e3, "shape": {"buffers": B, "components": K, "frames": T, "samples": n},
           "components_add_up_max_abs_err": err,
           "roofline": {"bound": "hbm", "achieved": nbytes / dd / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": nbytes / dd / 1e9 / PEAK_HBM_GBS, "bytes": nbytes, "traffic": None,
                        "other_view": {"TFLOP/s (vector FP64)": flops / dd / 1e12},
                        "per": "whole call (reciprocal V-hat pre-pass + the batched kernel)"},
           "device": device_line(ctx)}
    c.close()
    return res


def bench_line_configs(ctx, names=("c2", "c3", "c5")):
    """What bench.py appends to its JSON line as `configs`: the BASELINE configs other than the headline, timed by the same
    process right behind it (SURVEY 8(d) names c2 and c3 beside c4; c5 is the feature pipeline).  One compact entry per
    config -- time per unit, the roofline that bounds it with the algorithmic work per unit -- or {"error": ...}."""
    out = {}
    for name in names:
        t0 = time.perf_counter()
        try:
            if name == "c5":
                r = run_c5(ctx, False)
                e = {"workload": r["workload"], "ms": r.get("ms"), "value": r["value"], "unit": r["unit"],
                     "kernel_ms": r.get("kernel_ms"), "roofline": r.get("roofline"), "shape": r["shape"]}
            else:
                r = run_nmf_config(ctx, name, False)
                e = {"workload": r["workload"], "us_per_iteration": r["us_per_iteration"], "value": r["value"],
                     "unit": r["unit"], "iterations_timed": r["shape"]["iterations_timed"],
                     "roofline": r["roofline"], "kernel_ms_per_iteration": r["kernel_ms_per_iteration"],
                     "stft_frames_per_s": r["stft_frames_per_s"], "roofline_stft": r["roofline_stft"],
                     "schedule": r["schedule"], "shape": r["shape"]}
        except Exception as ex:  # a config that fails must not take the headline line with it
            e = {"error": f"{type(ex).__name__}: {ex}"}
        e["wall_s"] = time.perf_counter() - t0
        out[name] = e
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    with_cpu = "--no-cpu" not in sys.argv
    names = args or ["c1", "c2", "c4x1"]
    ctx = fluhip.Context(0)
    for name in names:
        if name == "c5":
            res = run_c5(ctx, with_cpu)
        elif name == "resynth":
            res = run_resynth(ctx)
        else:
            res = run_nmf_config(ctx, name, with_cpu)
        print(json.dumps(res), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
