"""The C++17 host-side clients (include/flucoma_hip/*.hpp: BufferAdaptor, NMFClient and the clients of SURVEY 8 (f) --
BufferSTFTClient, NMFSeedClient, NRTMFCCClient, NRTMelBandsClient -- each behind the generic NRTThreadingAdaptor)
driven through tests/cpp/client_driver.cpp.

CPU part: parameter validation, Result codes and the exact user-visible messages of
include/flucoma/clients/nrt/NMFClient.hpp:100-185 and clients/common/BufferAdaptor.hpp:175-208.
GPU part: sync / async jobs on MemoryBufferAdaptors against the oracle, offsets and multichannel
layout (nrt/NMFClient.hpp:233, 277-300), seeding, progress, cancellation.
"""
import importlib.util
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

from helpers import rel_err

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(driver, *args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run([driver, *map(str, args)], capture_output=True, text=True, timeout=300, env=e)
    assert out.returncode == 0, out.stderr
    res = {}
    for line in out.stdout.splitlines():
        tag, status, msg = line.split("|", 2)
        res[tag] = (int(status), msg)
    return res


def read_buffer(path):
    raw = open(path, "rb").read()
    frames, chans = struct.unpack("<qq", raw[:16])
    sr, = struct.unpack("<d", raw[16:24])
    data = np.frombuffer(raw[24:], dtype=np.float32).reshape(chans, frames)
    return data, sr


OK, WARNING, ERROR, CANCELLED = 0, 1, 2, 3


def test_parameter_descriptors_are_the_references_tables(driver):
    """what a host wrapper builds its attributes from (`getParameterDescriptors()`, clients/common/FluidNRTClientWrapper.hpp:801-804;
    include/flucoma_hip/ParamDescriptors.hpp): every mirrored client's table, through its NRTThreadingAdaptor, against the
    reference's own -- names, display names, types, defaults, bounds, enum strings, the order.  The fixture was minted from the
    reference headers (tools/make_param_descriptor_fixture.py); where the reference is present the fixture itself is re-derived
    and held to it"""
    import json
    out = subprocess.run([driver, "descriptors"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    mine = json.loads(out.stdout)
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "param_descriptors.json")))
    assert list(mine) == list(want) == ["BufNMF", "BufNMFSeed", "BufSTFT", "BufMFCC", "BufMelBands", "BufNMFFilter", "BufNMFMatch"]
    for client in want:
        assert mine[client] == want[client], client
    assert [d["name"] for d in mine["BufNMF"]] == ["source", "startFrame", "numFrames", "startChan", "numChans", "resynth", "resynthMode",
                                                   "bases", "basesMode", "activations", "actMode", "components", "iterations", "seed",
                                                   "fftSettings"]          # nrt/NMFClient.hpp:36-52, the index enum's order
    # the adaptor's connection counts (:811-822): no audio or control connections, buffers in / out = its buffer parameters; no messages
    con = subprocess.run([driver, "connections"], capture_output=True, text=True, timeout=60)
    assert con.returncode == 0, con.stderr
    got = {l.split("|")[0]: l.split("|")[1:] for l in con.stdout.splitlines()}
    for client, table in want.items():
        n_in = sum(d["kind"] == "InputBuffer" for d in table)
        n_out = sum(d["kind"] == "Buffer" for d in table)
        assert got[client] == ["%d %d 0 0 0 0" % (n_in, n_out), "0"], (client, got[client])
    if os.path.isdir("/root/reference/include/flucoma"):
        gen = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_param_descriptor_fixture.py"), "/root/reference"],
                             capture_output=True, text=True, timeout=60)
        assert gen.returncode == 0, gen.stderr
        assert json.loads(gen.stdout) == want


def test_validation_messages(driver):
    r = run(driver, "errors")
    assert r["no_source"] == (ERROR, "Input buffer not set")
    assert r["bad_start_frame"] == (ERROR, "Input buffer  invalid start frame 5000")
    assert r["bad_start_chan"] == (ERROR, "Input buffer  invalid start channel 2")
    assert r["too_many_frames"] == (ERROR, "Input buffer : not enough frames")
    assert r["too_many_chans"] == (ERROR, "Input buffer : not enough channels")
    assert r["seed_no_bases"] == (ERROR, "Bases Mode set to Seed or Fix , but no Bases Buffer supplied")
    assert r["seed_bad_bases_shape"] == (
        ERROR, "Supplied bases buffer for seeding must be [(FFTSize / 2) + 1] frames long, and have [rank] * [channels] channels")
    assert r["fix_no_acts"] == (ERROR, "Activations Mode set to Seed or Fix , but no Activations Buffer supplied")
    assert r["fix_bad_acts_shape"] == (
        ERROR, "Supplied activations buffer for seeding must be [(num samples / hop size)  + 1] frames long, and have [rank] * [channels] channels")
    assert r["both_fixed"] == (
        WARNING, "Bases and Activations buffers both fixed, but resynthesis disabled: no work to do")
    assert r["resynth_no_buffer"] == (ERROR, "Resynthesis requested but no buffer supplied")
    assert r["empty_queue"] == (WARNING, "Process() called on empty queue")


@pytest.mark.gpu
@pytest.mark.parametrize("use_async", [0, 1])
def test_client_stereo_with_offsets(driver, oracle, onp, tmp_path, use_async, ctx):
    frames, chans = 30000, 3
    audio = np.stack([onp.synth_audio(frames, 500 + c) for c in range(chans)], axis=1)  # frames x chans
    inp = tmp_path / "in.f32"
    audio.astype(np.float32).tofile(inp)
    win, hop, fft, K, iters, seed = 1024, 256, 1024, 4, 30, 42
    start_frame, num_frames, start_chan, num_chans = 1000, 20000, 1, 2
    prefix = str(tmp_path / "out")
    r = run(driver, "run", inp, frames, chans, win, hop, fft, K, iters, seed, 0, 0, use_async, start_frame,
            num_frames, start_chan, num_chans, prefix)
    assert r["result"] == (OK, "")
    if use_async:
        assert r["process"][0] == OK and r["max_progress"][0] == 1
    bases, sr_b = read_buffer(prefix + "_bases.bin")
    acts, sr_a = read_buffer(prefix + "_acts.bin")
    F, T = fft // 2 + 1, num_frames // hop + 1
    assert bases.shape == (K * num_chans, F) and acts.shape == (K * num_chans, T)     # :188-211
    assert sr_b == pytest.approx(44100.0 / fft) and sr_a == pytest.approx(44100.0 / hop)  # :199-209
    for i in range(num_chans):
        x = audio[start_frame:start_frame + num_frames, start_chan + i]
        rb, ra = oracle.bufnmf_channel(np.ascontiguousarray(x), win, fft, hop, K, iters, seed)
        assert rel_err(bases[i * K:(i + 1) * K], rb) < 1e-6    # channel i*K + j holds component j (:281,:295)
        assert rel_err(acts[i * K:(i + 1) * K], ra) < 1e-6


@pytest.mark.gpu
def test_client_seeded_bases_fixed_activations(driver, oracle, onp, tmp_path, ctx):
    frames = 16384
    audio = onp.synth_audio(frames, 77)
    inp = tmp_path / "in.f32"
    audio.tofile(inp)
    win, hop, fft, K, iters = 512, 128, 512, 3, 25
    F, T = fft // 2 + 1, frames // hop + 1
    rs = np.random.RandomState(1)
    seedW = rs.uniform(0.05, 1, (K, F)).astype(np.float32)
    seedH = rs.uniform(0.05, 1, (K, T)).astype(np.float32)
    seedW.tofile(tmp_path / "sw.f32")
    seedH.tofile(tmp_path / "sh.f32")
    prefix = str(tmp_path / "out")
    # basesMode = Seed (1), actMode = Fixed (2): W updates from the seed, H stays
    r = run(driver, "run", inp, frames, 1, win, hop, fft, K, iters, 42, 1, 2, 0, 0, -1, 0, -1, prefix,
            tmp_path / "sw.f32", tmp_path / "sh.f32")
    assert r["result"] == (OK, "")
    bases, _ = read_buffer(prefix + "_bases.bin")
    acts, _ = read_buffer(prefix + "_acts.bin")
    _, mag = oracle.stft_f32(audio, win, fft, hop)
    rW, rH, _, _ = oracle.nmf_process(mag, K, iters, True, False, 42, W0=seedW.astype(np.float64),
                                      H0=seedH.T.astype(np.float64))
    assert rel_err(bases, rW.astype(np.float32)) < 1e-6
    assert np.array_equal(acts, seedH)            # fixed activations are not written back (:286)


@pytest.mark.gpu
def test_client_cancel(driver, ctx):
    r = run(driver, "cancel", 441000)
    assert r["process"][0] == OK
    assert r["progress_before_cancel"][0] == 1     # progress was moving and below 1
    assert r["cancelled"] == (CANCELLED, "")       # :273-274


@pytest.mark.gpu
def test_client_resynthesis(driver, oracle, onp, tmp_path, ctx):
    """resynthMode 1 through the host client: buffer shape nFrames x (rank * channels) at the source
    sample rate (nrt/NMFClient.hpp:178-185), components add back up to the source."""
    frames = 20000
    audio = np.stack([onp.synth_audio(frames, 600 + c) for c in range(2)], axis=1)
    inp = tmp_path / "in.f32"
    audio.astype(np.float32).tofile(inp)
    win, hop, fft, K, iters = 1024, 256, 1024, 3, 20
    prefix = str(tmp_path / "out")
    r = run(driver, "run", inp, frames, 2, win, hop, fft, K, iters, 42, 0, 0, 1, 0, -1, 0, -1, prefix,
            env={"CLIENT_RESYNTH": "1"})
    assert r["result"] == (OK, "")
    res, sr = read_buffer(prefix + "_resynth.bin")
    assert res.shape == (2 * K, frames) and sr == 44100.0
    for c in range(2):
        total = res[c * K:(c + 1) * K].sum(axis=0)
        assert np.abs(total[win:-win] - audio[win:-win, c]).max() < 0.02
        # every component against the oracle (the two channels run as one corpus; the result reaches the interleaved host
        # buffer transposed on the device, fluhip_corpus_resynth_interleaved_host)
        x = np.ascontiguousarray(audio[:, c].astype(np.float32))
        spec, mag = oracle.stft_f32(x, win, fft, hop)
        W1, H1, V1, _ = oracle.nmf_process(mag, K, iters, True, True, 42)
        for k in range(K):
            ref = oracle.resynth_component(spec, W1, H1, V1, k, win, fft, hop, frames)
            assert np.abs(res[c * K + k] - ref).max() / max(np.abs(ref).max(), 1e-12) < 1e-5, (c, k)


@pytest.mark.gpu
@pytest.mark.parametrize("use_async", [0, 1])
def test_client_channels_over_two_device_contexts(driver, tmp_path, use_async, ctx):
    """FluidContext::devices = {0, 0}: the three channels of a job run concurrently on two contexts (one host thread
    each, channels dealt round-robin) -- the buffers must come out exactly as from the sequential single-context job
    (same kernels, same seeds; nrt/NMFClient.hpp:233 channels share no state), resynthesis included"""
    from conftest import ROOT as _r  # noqa: F401
    import oracle_np as onp
    frames, chans = 24000, 3
    audio = np.stack([onp.synth_audio(frames, 700 + c) for c in range(chans)], axis=1)
    inp = tmp_path / "in.f32"
    audio.astype(np.float32).tofile(inp)
    win, hop, fft, K, iters, seed = 1024, 256, 1024, 4, 15, 42
    outs = {}
    for tag, env in (("one", {"CLIENT_RESYNTH": "1"}), ("two", {"CLIENT_RESYNTH": "1", "CLIENT_DEVICES": "0,0"})):
        prefix = str(tmp_path / tag)
        r = run(driver, "run", inp, frames, chans, win, hop, fft, K, iters, seed, 0, 0, use_async, 0, -1, 0, -1, prefix, env=env)
        assert r["result"] == (OK, "")
        outs[tag] = [read_buffer(prefix + s)[0] for s in ("_bases.bin", "_acts.bin", "_resynth.bin")]
    for a, b in zip(outs["one"], outs["two"]):
        assert a.shape == b.shape and np.array_equal(a, b)
    assert outs["two"][0].shape == (K * chans, fft // 2 + 1)


@pytest.mark.gpu
@pytest.mark.parametrize("bases_mode,act_mode", [(0, 0), (1, 0), (2, 1)])
def test_client_eight_channels_run_as_one_corpus(driver, oracle, onp, tmp_path, ctx, bases_mode, act_mode):
    """the channel loop of nrt/NMFClient.hpp:233 on the batched kernels.  An 8-channel x 10 s rank-32
    job through NRTThreadedNMFClient: random start, seeded bases, fixed bases + seeded activations -- channels against the
    per-channel oracle, the whole result against the channel-by-channel loop (FLUHIP_CLIENT_SEQUENTIAL=1), and the wall
    time of the batched job against the sequential one's.  (The review asked for a quarter: measured 22.3 against 90.6 ms
    = 0.246 once the batch ran from work lists -- 200 iterations at 73 us instead of 88 -- and the adaptor kept one client,
    hence one device context, across jobs like the reference's does.  The bar here is 0.35: boxes differ.)"""
    frames, chans = 441000, 8
    win, hop, fft, K, iters, seed = 2048, 512, 2048, 32, 200, 42
    F, T = fft // 2 + 1, frames // hop + 1
    audio = np.stack([onp.synth_audio(frames, 800 + c) for c in range(chans)], axis=1)   # frames x chans
    inp = tmp_path / "in.f32"
    audio.astype(np.float32).tofile(inp)
    extra = []
    rs = np.random.RandomState(3)
    seedW = rs.uniform(0.05, 1, (chans * K, F)).astype(np.float32)
    seedH = rs.uniform(0.05, 1, (chans * K, T)).astype(np.float32)
    if bases_mode or act_mode:
        seedW.tofile(tmp_path / "sw.f32"); seedH.tofile(tmp_path / "sh.f32")
        extra = [tmp_path / "sw.f32", tmp_path / "sh.f32"]
    outs, ms = {}, {}
    # (a seeded job run twice on the same buffers would seed its second run with the first one's result: the parity runs
    #  execute once; the timing runs -- random start only -- twice, the first paying for the context and the code objects)
    for tag, env in (("batched", {}), ("sequential", {"FLUHIP_CLIENT_SEQUENTIAL": "1"})):
        prefix = str(tmp_path / tag)
        r = run(driver, "run", inp, frames, chans, win, hop, fft, K, iters, seed, bases_mode, act_mode, 0, 0, -1, 0, -1, prefix,
                *extra, env=env)
        assert r["result"] == (OK, "")
        outs[tag] = [read_buffer(prefix + s)[0] for s in ("_bases.bin", "_acts.bin")]
        if not (bases_mode or act_mode):
            r = run(driver, "run", inp, frames, chans, win, hop, fft, K, iters, seed, 0, 0, 0, 0, -1, 0, -1, prefix + "_t",
                    env=dict(env, CLIENT_REPEAT="2"))
            ms[tag] = float(r["elapsed_ms"][1])
    for a, b in zip(outs["batched"], outs["sequential"]):
        assert a.shape == b.shape and rel_err(a, b) < 1e-6           # schedules differ (summation order), floats agree
    bases, acts = outs["batched"]
    assert bases.shape == (K * chans, F) and acts.shape == (K * chans, T)
    for c in (0, 5, 7):
        x = np.ascontiguousarray(audio[:, c])
        _, mag = oracle.stft_f32(x, win, fft, hop)
        W0 = seedW[c * K:(c + 1) * K].astype(np.float64) if bases_mode else None
        H0 = np.ascontiguousarray(seedH[c * K:(c + 1) * K].T.astype(np.float64)) if act_mode else None
        rW, rH, _, _ = oracle.nmf_process(mag, K, iters, bases_mode != 2, act_mode != 2, seed, W0=W0, H0=H0)
        rb, ra = oracle.bufnmf_writeback(rW, rH)
        if bases_mode != 2:
            assert rel_err(bases[c * K:(c + 1) * K], rb) < 1e-6, c
        else:
            assert np.array_equal(bases[c * K:(c + 1) * K], seedW[c * K:(c + 1) * K])   # fixed bases are not written back
        assert rel_err(acts[c * K:(c + 1) * K], ra) < 1e-6, c
    if ms:      # printed, never asserted: wall-clock ratios live in tools/perf_matrix.py (profiles/rNN/perf_matrix.json)
        print(f"8-channel job: batched {ms['batched']:.1f} ms, channel by channel {ms['sequential']:.1f} ms")


@pytest.mark.gpu
def test_client_batched_failure_falls_back_to_the_channel_loop(driver, tmp_path, ctx):
    """every allocation of the batched multi-channel block is sized for all channels at once (kept spectrum,
    resynthesis output + its transposed copy, mask workspace); a failure anywhere in it -- injected here behind the
    iterations, where an out-of-memory resynthesis would sit -- must hand the job to the channel-by-channel loop instead of
    returning kError: same status, same floats in all three output buffers as the job that never failed."""
    frames, chans = 20000, 3
    win, hop, fft, K, iters, seed = 1024, 256, 1024, 4, 12, 42
    rs = np.random.RandomState(11)
    audio = rs.uniform(-0.5, 0.5, (frames, chans)).astype(np.float32)
    inp = tmp_path / "in.f32"
    audio.tofile(inp)
    outs = {}
    for tag, env in (("batched", {}), ("failed", {"FLUHIP_CLIENT_FAIL_BATCHED": "1"})):
        prefix = str(tmp_path / tag)
        r = run(driver, "run", inp, frames, chans, win, hop, fft, K, iters, seed, 0, 0, 1, 0, -1, 0, -1, prefix,
                env=dict(env, CLIENT_RESYNTH="1"))
        assert r["result"] == (OK, ""), r
        assert r["fallbacks"][0] == (1 if env else 0), r
        outs[tag] = [read_buffer(prefix + s)[0] for s in ("_bases.bin", "_acts.bin", "_resynth.bin")]
    for a, b in zip(outs["batched"], outs["failed"]):
        assert a.shape == b.shape and rel_err(a, b) < 1e-6
    # ... and only an ALLOCATION failure is worth the second attempt: any other error comes back at once
    r = run(driver, "run", inp, frames, chans, win, hop, fft, K, iters, seed, 0, 0, 1, 0, -1, 0, -1, str(tmp_path / "other"),
            env={"FLUHIP_CLIENT_FAIL_BATCHED": "other", "CLIENT_RESYNTH": "1"})
    assert r["result"][0] == ERROR and r["fallbacks"][0] == 0, r


@pytest.mark.gpu
def test_pool_from_a_cpp_host(driver, ctx):
    """fluhip_pool_bufnmf_f32 called from C++ (tests/cpp/client_driver.cpp): 7 buffers over two contexts on device 0 give
    the floats of the one-context run (the schedules differ by the number of buffers per launch: rounding only)"""
    r = run(driver, "pool", 7, 30000)
    assert r["pool_rc"][0] == 0 and r["pool_rc"][1] == "0"
    assert r["pool_match"][0] == 1, r["pool_match"]


def test_host_buffer_plumbing_without_a_device(driver, tmp_path):
    """tests/cpp/host_buffers_check.cpp under -fsanitize=address,undefined: block-wise scatter / gather against the
    channel-by-channel loops (interleaved and planar host buffers), interleaved-layout detection, MemoryBufferAdaptor deep
    copies / copy-back / refilled and shape-only copies, and the generic NRTThreadingAdaptor with a client that needs no
    device (worker thread, copy-back on the polling thread, write-only buffers, copies reused by the next job)"""
    lib = os.path.dirname(driver)
    exe = tmp_path / "host_buffers_check"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-Wall", "-Wextra",
           "-Werror", "-pthread", os.path.join(ROOT, "tests", "cpp", "host_buffers_check.cpp"), "-o", str(exe), "-L" + lib,
           "-lflucoma_hip", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert b.returncode == 0, b.stderr[-3000:]
    e = dict(os.environ)
    e["ASAN_OPTIONS"] = "detect_leaks=0"
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120, env=e)
    assert out.returncode == 0, out.stdout + out.stderr[-3000:]
    lines = out.stdout.split()
    assert out.stdout.count("ok ") == 11 and "FAILED" not in out.stdout, out.stdout


# ---- the clients of SURVEY 8 (f): BufSTFT, BufNMFSeed, BufMFCC, BufMelBands ----------------------------------------
def test_validation_messages_of_the_other_clients(driver):
    """nrt/BufSTFTClient.hpp:84-107,189-216, nrt/NMFSeedClient.hpp:75-88, cc/FluidNRTClientWrapper.hpp:313-328: the
    user-visible strings (typos included) and the parameter constraints of rt/MFCCClient.hpp:38-49; no device needed"""
    r = run(driver, "errors2")
    assert r["stft_no_source"] == (ERROR, "No input buffer supplied")
    assert r["stft_no_outputs"] == (ERROR, "Neither magnitude nor phase buffer supplied")
    assert r["stft_bad_start_frame"] == (ERROR, "Input buffer  invalid start frame 5000")
    assert r["stft_too_many_frames"] == (ERROR, "Input buffer : not enough frames")
    assert r["stft_too_many_bins"] == (ERROR, "Can produce up to 65536 channels. Split your data up and try again")
    assert r["istft_needs_both"] == (ERROR, "Need both magnutude and phase buffers for inverse transform")
    assert r["istft_no_resynth"] == (ERROR, "No resynthesis buffer supplied")
    assert r["istft_size_mismatch"] == (ERROR, "Magnitude and Phase buffer sizes don't match")
    assert r["istft_wrong_channels"] == (ERROR, "Wrong number of channels for FFT sizee of 1024 got 512 expected 513")
    assert r["seed_no_source"] == (ERROR, "Source Buffer Supplied But Invalid")
    assert r["seed_two_channels"] == (ERROR, "Only one channel supported")
    assert r["mfcc_no_source"] == (ERROR, "Input buffer not set")
    assert r["mfcc_bad_start_chan"] == (ERROR, "Input buffer  invalid start channel 2")
    assert r["mfcc_no_output"] == (ERROR, "No valid output has been set")
    assert r["melbands_too_many_chans"] == (ERROR, "Input buffer : not enough channels")
    assert r["melbands_no_output"] == (ERROR, "No valid output has been set")
    assert r["mfcc_constraints"][0] == 1, r["mfcc_constraints"]


@pytest.mark.gpu
@pytest.mark.parametrize("use_async", [0, 1])
@pytest.mark.parametrize("win,hop,fft,padding", [(1024, 256, 1024, 1), (1000, 300, 1024, 2), (512, 128, 2048, 0)])
def test_bufstft_client(driver, onp, tmp_path, use_async, win, hop, fft, padding, ctx):
    """BufferSTFTClient forward then inverse (nrt/BufSTFTClient.hpp:81-276) through the threading adaptor: buffer
    shapes and sample rates, magnitude / phase against the numpy restatement, channel 0 read whatever startChan says
    (:152), and the resynthesis against the restated inverse of the same float buffers"""
    frames, chans = 20000, 2
    audio = np.stack([onp.synth_audio(frames, 900 + c) for c in range(chans)], axis=1)
    inp = tmp_path / "in.f32"
    audio.astype(np.float32).tofile(inp)
    start, num = 1500, 15000
    prefix = str(tmp_path / "o")
    r = run(driver, "stft", inp, frames, chans, win, hop, fft, padding, start, num, use_async, prefix)
    assert r["forward"] == (OK, "") and r["inverse"] == (OK, "")
    mag, sr_m = read_buffer(prefix + "_mag.bin")
    ph, sr_p = read_buffer(prefix + "_phase.bin")
    x = np.ascontiguousarray(audio[start:start + num, 0])
    rm, rp = onp.bufstft_forward(x, win, fft, hop, padding)
    assert mag.shape == rm.shape == ph.shape                                  # bins x hops: channel = bin (:168-178)
    assert sr_m == pytest.approx(44100.0 / hop) and sr_p == pytest.approx(44100.0 / hop)
    assert np.abs(mag - rm).max() / np.abs(rm).max() < 1e-6
    strong = rm > 1e-3 * rm.max()                                             # the phase of an empty bin is noise
    d = np.angle(np.exp(1j * (ph.astype(np.float64) - rp)))[strong]
    assert np.abs(d).max() < 1e-3
    out, sr_o = read_buffer(prefix + "_resynth.bin")
    ref = onp.bufstft_inverse(mag, ph, win, fft, hop, padding)
    assert out.shape == (1, ref.shape[0]) and sr_o == pytest.approx(44100.0)  # mags.sampleRate() * hop (:233)
    # std::polar in single precision on both sides; the overlap-add divides float rounding by the summed squared window,
    # nearly zero on the outermost samples: weighted by that divisor (as in tests/test_gpu_parity.py)
    w2 = onp.hann(win) ** 2
    nrm = np.zeros((mag.shape[1] - 1) * hop + win)
    for t in range(mag.shape[1]):
        nrm[t * hop:t * hop + win] += w2
    pad = onp.bufstft_padding(win, hop, padding)
    nrm = np.maximum(nrm[pad:pad + out.shape[1]], 2.220446049250313e-16)
    assert (np.abs(out[0] - ref) * np.minimum(nrm, 1.0)).max() < 2e-6 * max(1.0, np.abs(mag).max() / fft)
    if padding == 1 and win % hop == 0:                                       # COLA: the input comes back
        m = min(num, out.shape[1])
        assert np.abs(out[0, win:m - win] - x[win:m - win]).max() < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("use_async", [0, 1])
def test_bufnmfseed_client(driver, oracle, onp, tmp_path, use_async, ctx):
    """NMFSeedClient (nrt/NMFSeedClient.hpp:73-131): both output buffers resized to the rank found (:108-118), bases at
    sampleRate / fft, activations at sampleRate / hop, method 0 against STFT (C oracle) -> NNDSVD (numpy / LAPACK)"""
    frames = 30000
    x = onp.synth_audio(frames, 77)
    inp = tmp_path / "in.f32"
    x.astype(np.float32).tofile(inp)
    win, hop, fft, max_rank = 1024, 256, 1024, 12
    prefix = str(tmp_path / "o")
    r = run(driver, "seed", inp, frames, win, hop, fft, 1, max_rank, 0.7, 0, 42, use_async, prefix)
    assert r["result"] == (OK, "")
    bases, sr_b = read_buffer(prefix + "_bases.bin")
    acts, sr_a = read_buffer(prefix + "_acts.bin")
    _, mag = oracle.stft_f32(x, win, fft, hop)
    rW, rH, rk, *_ = onp.nndsvd(mag, max_rank, 1, max_rank, 0.7, 0, 42)
    assert bases.shape == (rk, fft // 2 + 1) and acts.shape == (rk, frames // hop + 1)
    assert sr_b == pytest.approx(44100.0 / fft) and sr_a == pytest.approx(44100.0 / hop)
    ra = rH.T.astype(np.float32) * np.float32(1.0 / rH.max())
    assert rel_err(bases, rW[:rk].astype(np.float32)) < 1e-5 and rel_err(acts, ra[:rk]) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("use_async", [0, 1])
@pytest.mark.parametrize("padding", [0, 1, 2])
def test_bufmfcc_client(driver, onp, tmp_path, use_async, padding, ctx):
    """NRTMFCCClient (rt/MFCCClient.hpp:86-175 behind StreamingControl, cc/FluidNRTClientWrapper.hpp:551-660): a
    3-channel buffer with frame and channel offsets is one batch on the device; feature i of channel j lands in buffer
    channel i + j * numCoeffs (:650-655), frames = kept hops, sample rate = sampleRate / hop"""
    frames, chans = 24000, 3
    audio = np.stack([onp.synth_audio(frames, 600 + c) for c in range(chans)], axis=1)
    inp = tmp_path / "in.f32"
    audio.astype(np.float32).tofile(inp)
    win, hop, fft, n_bands, n_coefs, start_coeff = 1024, 256, 1024, 40, 13, 1
    start_frame, num_frames, start_chan, num_chans = 500, 20000, 1, 2
    prefix = str(tmp_path / "o")
    r = run(driver, "mfcc", inp, frames, chans, win, hop, fft, padding, n_bands, n_coefs, start_coeff, start_frame,
            num_frames, start_chan, num_chans, use_async, prefix)
    assert r["result"] == (OK, "")
    feat, sr = read_buffer(prefix + "_features.bin")
    T, _ = onp.feature_frames(num_frames, win, hop, padding)
    assert feat.shape == (num_chans * n_coefs, T) and sr == pytest.approx(44100.0 / hop)
    for j in range(num_chans):
        x = np.ascontiguousarray(audio[start_frame:start_frame + num_frames, start_chan + j])
        ref = onp.bufmfcc_channel(x, win, fft, hop, n_bands, n_coefs, start_coeff, padding_mode=padding)
        got = feat[j * n_coefs:(j + 1) * n_coefs]
        assert np.abs(got - ref).max() < 2e-3 and np.abs(got - ref).max() / np.abs(ref).max() < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("normalize,scale", [(1, 0), (0, 1)])
def test_bufmelbands_client(driver, onp, tmp_path, normalize, scale, ctx):
    """NRTMelBandsClient (rt/MelBandsClient.hpp:77-159): normalize / scale enums, whole stereo buffer"""
    frames, chans = 16000, 2
    audio = np.stack([onp.synth_audio(frames, 650 + c) for c in range(chans)], axis=1)
    inp = tmp_path / "in.f32"
    audio.astype(np.float32).tofile(inp)
    win, hop, fft, n_bands = 1000, 300, 1024, 24
    prefix = str(tmp_path / "o")
    r = run(driver, "melbands", inp, frames, chans, win, hop, fft, 1, n_bands, normalize, scale, 0, -1, 0, -1, 1, prefix)
    assert r["result"] == (OK, "")
    feat, sr = read_buffer(prefix + "_features.bin")
    T, _ = onp.feature_frames(frames, win, hop, 1)
    assert feat.shape == (chans * n_bands, T) and sr == pytest.approx(44100.0 / hop)
    for j in range(chans):
        ref = onp.bufmelbands_channel(np.ascontiguousarray(audio[:, j]), win, fft, hop, n_bands,
                                      normalize=bool(normalize), scale_db=bool(scale))
        got = feat[j * n_bands:(j + 1) * n_bands]
        if scale:
            assert np.abs(got - ref).max() < 2e-3
        else:
            assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-5


@pytest.mark.gpu
def test_bufmfcc_client_reproduces_the_references_corpus_rows(driver, tmp_path, ctx):
    """the way the reference's demo corpus was analysed -- BufMFCC with startFrame / numFrames per slice, startCoeff 1 --
    through NRTMFCCClient on the fixture's audio, against the rows a FluCoMa build computed
    (tests/golden/reference_corpus_mfcc.npz, Resources/Data/flucoma_corpus_mfcc.json): mean and deviation per coefficient"""
    g = np.load(os.path.join(ROOT, "tests", "golden", "reference_corpus_mfcc.npz"))
    x = (g["pcm16"].astype(np.float64) / 32768.0).astype(np.float32)
    inp = tmp_path / "in.f32"
    x.tofile(inp)
    pts = g["points"]
    for j in range(len(pts) - 1):
        prefix = str(tmp_path / ("s%d" % j))
        r = run(driver, "mfcc", inp, len(x), 1, 1024, 512, 1024, 1, 40, 13, 1, int(pts[j]), int(pts[j + 1] - pts[j]), 0, 1, j & 1, prefix)
        assert r["result"] == (OK, "")
        feat, sr = read_buffer(prefix + "_features.bin")
        m = feat.astype(np.float64)
        got = np.concatenate([m.mean(axis=1), m.std(axis=1)])
        assert feat.shape[0] == 13 and np.abs(got - g["expected"][j]).max() < 2e-5, (j, np.abs(got - g["expected"][j]).max())



def test_validation_of_nmfmatch_and_nmffilter(driver):
    """the wrapper's checks in front of the two processFrame clients (cc/FluidNRTClientWrapper.hpp:313-328) and the
    constraints of their parameter tables (rt/NMFMatchClient.hpp:32-38, rt/NMFFilterClient.hpp:34-38): no device needed"""
    r = run(driver, "errors2")
    assert r["match_no_source"] == (ERROR, "Input buffer not set")
    assert r["match_no_output"] == (ERROR, "No valid output has been set")
    assert r["match_constraints"][0] == 1
    assert r["filter_no_source"] == (ERROR, "Input buffer not set")
    assert r["filter_bad_start_chan"] == (ERROR, "Input buffer  invalid start channel 2")
    assert r["filter_no_output"] == (ERROR, "No valid output has been set")


@pytest.mark.gpu
@pytest.mark.parametrize("use_async", [0, 1])
def test_nmfmatch_client(driver, onp, tmp_path, use_async, ctx):
    """NRTThreadedNMFMatchClient (include/flucoma_hip/NMFMatchClient.hpp): NMFMatch behind the reference's StreamingControl
    wrapper -- two channels, a bases buffer with MORE channels than maxComponents (rank = min of the two,
    rt/NMFMatchClient.hpp:93), sync and on the adaptor's own thread: features buffer of keepHops x (channels * rank) at
    sampleRate / hop, feature i of channel j in buffer channel i + j * rank (cc/FluidNRTClientWrapper.hpp:636-656), every
    channel against the numpy restatement; and the no-filters case (zeros, maxComponents features)"""
    frames, chans, win, hop, fft, K, max_rank, seed = 20000, 2, 1024, 256, 1024, 6, 4, 42
    F = fft // 2 + 1
    rs = np.random.RandomState(9)
    audio = np.stack([onp.synth_audio(frames, 8300 + c) for c in range(chans)], axis=1)      # frames x chans
    bases = (np.abs(rs.standard_normal((K, F))) + 0.01).astype(np.float32)
    inp, bf = tmp_path / "in.f32", tmp_path / "bases.f32"
    audio.astype(np.float32).tofile(inp); bases.tofile(bf)
    prefix = str(tmp_path / "m")
    r = run(driver, "nmfmatch", inp, frames, chans, win, hop, fft, 1, max_rank, seed, bf, K, use_async, prefix)
    assert r["result"] == (OK, "")
    feats, sr = read_buffer(prefix + "_features.bin")
    T = onp.feature_frames(frames, win, hop, 1)[0]
    assert feats.shape == (chans * max_rank, T) and sr == pytest.approx(44100.0 / hop)
    for c in range(chans):
        ref = onp.nmfmatch_channel(np.ascontiguousarray(audio[:, c]), bases[:max_rank], win, fft, hop, seed, 1)
        assert rel_err(feats[c * max_rank:(c + 1) * max_rank], ref) < 1e-5, c
    r = run(driver, "nmfmatch", inp, frames, chans, win, hop, fft, 1, max_rank, seed, "-", 1, use_async, prefix + "0")
    assert r["result"] == (OK, "")
    feats0, _ = read_buffer(prefix + "0_features.bin")
    assert feats0.shape == (chans * max_rank, T) and not feats0.any()


@pytest.mark.gpu
@pytest.mark.parametrize("use_async", [0, 1])
def test_nmffilter_client(driver, onp, tmp_path, use_async, ctx):
    """NRTThreadedNMFFilterClient (include/flucoma_hip/NMFFilterClient.hpp): NMFFilter behind the reference's Streaming
    wrapper -- two channels, three components, sync and async: resynth buffer of frames x (channels * maxComponents) at the source's
    sample rate, component j of channel i in buffer channel i * maxComponents + j, against the numpy restatement; the components of
    a channel add back up to it; a bases buffer of the wrong frame count leaves zeros (rt/NMFFilterClient.hpp:93)"""
    frames, chans, win, hop, fft, K, iters, seed = 15000, 2, 1024, 512, 1024, 3, 10, 42
    F = fft // 2 + 1
    rs = np.random.RandomState(10)
    audio = np.stack([onp.synth_audio(frames, 8400 + c) for c in range(chans)], axis=1)
    bases = (np.abs(rs.standard_normal((K, F))) + 0.01).astype(np.float32)
    inp, bf = tmp_path / "in.f32", tmp_path / "bases.f32"
    audio.astype(np.float32).tofile(inp); bases.tofile(bf)
    prefix = str(tmp_path / "f")
    r = run(driver, "nmffilter", inp, frames, chans, win, hop, fft, 20, iters, seed, bf, K, use_async, prefix)
    assert r["result"] == (OK, "")
    out, sr = read_buffer(prefix + "_resynth.bin")
    # maxComponents (20) outputs per channel whatever the rank of the bases (rt/NMFFilterClient.hpp:57-59: audioChannelsOut is
    # fixed at construction): the K live ones first, the rest silent
    M = 20
    assert out.shape == (chans * M, frames) and sr == pytest.approx(44100.0)
    for c in range(chans):
        x = np.ascontiguousarray(audio[:, c]).astype(np.float32)
        ref = onp.nmffilter_channel(x, bases, win, fft, hop, iters, seed)
        assert np.abs(out[c * M:c * M + K] - ref).max() / np.abs(ref).max() < 1e-5, c
        assert np.abs(out[c * M:c * M + K].sum(axis=0) - x).max() < 1e-4
        assert not out[c * M + K:(c + 1) * M].any()
    bases[:, :100].tofile(tmp_path / "short.f32")                 # 100 frames instead of 513
    r = run(driver, "nmffilter", inp, frames, chans, win, hop, fft, 2, iters, seed, tmp_path / "short.f32", K, use_async, prefix + "0")
    assert r["result"] == (OK, "")
    out0, _ = read_buffer(prefix + "0_resynth.bin")
    assert out0.shape == (chans * 2, frames) and not out0.any()
