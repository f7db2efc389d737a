"""Timing of the other BASELINE configs (c1, c2, c3) through the C ABI: these are parity-test shapes,
not bench.py lines, but their per-iteration cost shows how the single-buffer (split-R) path behaves.
    python tools/bench_configs.py [c1 c2 c3]
"""
import os
import sys
import time

import numpy as np

try:  # torch first when it is there: the two then share one HIP runtime (config5's device-resident variant)
    import torch  # noqa: F401
except ImportError:
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flucoma-core_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import fluhip  # noqa: E402
import oracle_np  # noqa: E402

CONFIGS = {
    "c1": dict(n=453932, win=1024, fft=1024, hop=512, K=3, iters=50),
    "c2": dict(n=2646000, win=2048, fft=2048, hop=512, K=16, iters=200),
    "c3": dict(n=26460000, win=4096, fft=4096, hop=1024, K=128, iters=20),  # 1 channel, 20 of 500 iterations
    "c4x1": dict(n=441000, win=2048, fft=2048, hop=512, K=32, iters=200),
}


def main():
    names = [a for a in sys.argv[1:] if a != "c5"] or ([] if "c5" in sys.argv[1:] else ["c1", "c2", "c4x1"])
    ctx = fluhip.Context(0)
    for name in names:
        c = CONFIGS[name]
        # tile a 10 s synthetic clip to the requested length (content is irrelevant for timing)
        base = oracle_np.synth_audio(min(c["n"], 441000), 1000)
        x = np.tile(base, c["n"] // len(base) + 1)[:c["n"]].copy()
        cor = fluhip.Corpus(ctx, 1, c["n"], c["win"], c["fft"], c["hop"], c["K"])
        cor.set_audio(x[None, :])
        cor.stft(); ctx.synchronize()
        t0 = time.perf_counter(); cor.stft(); ctx.synchronize(); t_stft = time.perf_counter() - t0
        cor.nmf(2, seed=42); ctx.synchronize()
        # per-iteration cost = slope between two iteration counts (the host-side RNG init of the factors,
        # util/EigenRandom.hpp semantics, is a fixed cost per call: ~12 ms for c3's 3.6 M draws)
        n1 = max(2, c["iters"] // 4)
        t0 = time.perf_counter(); cor.nmf(n1, seed=42); ctx.synchronize(); t1 = time.perf_counter() - t0
        t0 = time.perf_counter(); cor.nmf(c["iters"], seed=42); ctx.synchronize(); t_nmf = time.perf_counter() - t0
        per_it = (t_nmf - t1) / (c["iters"] - n1)
        T, F, K = cor.T, cor.F, c["K"]
        flop = 8.0 * F * T * K
        print(f"{name}: T={T} F={F} K={K}  stft {t_stft*1e3:.2f} ms ({T/t_stft/1e6:.2f} Mframes/s)  "
              f"nmf {c['iters']} it {t_nmf*1e3:.1f} ms; {per_it*1e6:.1f} us/iteration "
              f"({flop/per_it/1e12:.1f} TF algorithmic), fixed {max(t_nmf - per_it*c['iters'], 0)*1e3:.1f} ms  "
              f"device {cor.device_bytes()/1e6:.0f} MB")
        cor.close()


if __name__ == "__main__":
    main()


def config5():
    """BASELINE config 5: STFT -> MelBands(40) -> MFCC(13) over 8192 x 2 s slices (host buffers in/out)."""
    import oracle_c
    ctx = fluhip.Context(0)
    count, n = 8192, 88200
    base = np.stack([oracle_np.synth_audio(n, 1000 + b) for b in range(64)])
    audio = np.tile(base, (count // 64, 1))
    ctx.bufmfcc(audio[:64], 1024, 1024, 512)
    t0 = time.perf_counter(); out = ctx.bufmfcc(audio, 1024, 1024, 512); dt = time.perf_counter() - t0
    frames = out.shape[0] * out.shape[2]
    o = oracle_c.get("native")
    t0 = time.perf_counter()
    for b in range(8):
        o.bufmfcc_channel(audio[b], 1024, 1024, 512)
    cpu = (time.perf_counter() - t0) / 8
    print(f"c5: {count} slices x {out.shape[2]} frames: {dt*1e3:.1f} ms incl. PCIe ({frames/dt/1e6:.1f} Mframes/s); "
          f"CPU oracle {cpu*1e3:.2f} ms per slice ({out.shape[2]/cpu/1e3:.1f} kframes/s, 1 core) -> {cpu*count/dt:.0f}x")
    try:  # the same with input and output resident in HBM (what a device-side pipeline would see)
        import ctypes
        if torch is None:
            raise ImportError
        a_dev = torch.from_numpy(audio).cuda()
        o_dev = torch.empty(out.shape, dtype=torch.float32, device="cuda")
        Tr = ctypes.c_int64(0)
        def run():
            rc = ctx.lib.fluhip_bufmfcc_f32(ctx.h, ctypes.cast(a_dev.data_ptr(), ctypes.POINTER(ctypes.c_float)), count, n,
                                            1024, 1024, 512, 40, 13, 0, 20.0, 20000.0, 44100.0,
                                            ctypes.cast(o_dev.data_ptr(), ctypes.POINTER(ctypes.c_float)), ctypes.byref(Tr))
            assert rc == 0
        run(); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(); torch.cuda.synchronize(); dd = time.perf_counter() - t0
        assert np.array_equal(o_dev.cpu().numpy(), out)
        print(f"c5 with audio and features resident in HBM: {dd*1e3:.1f} ms ({frames/dd/1e6:.0f} Mframes/s) -> {cpu*count/dd:.0f}x one CPU core")
    except ImportError:
        pass


if __name__ == "__main__" and "c5" in sys.argv[1:]:
    config5()
