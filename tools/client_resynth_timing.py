"""wall time of an 8-channel x 10 s rank-32 BufNMF job through the C++ client (NRTThreadedNMFClient, synchronous), without and
with the resynthesis output (8 x 32 x 441 000 floats through the host buffers): python tools/client_resynth_timing.py"""
import os, sys, subprocess, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flucoma-core_amd"))
import synth
frames, chans = 441000, 8
audio = np.stack([synth.synth_audio(frames, 500 + c) for c in range(chans)], axis=1).astype(np.float32)
d = tempfile.mkdtemp(); inp = os.path.join(d, "in.f32"); audio.tofile(inp)
drv = os.path.join(ROOT, "flucoma-core_amd", "lib", "client_driver")
for env in ({}, {"CLIENT_RESYNTH": "1"}):
    e = dict(os.environ); e.update(env); e["CLIENT_REPEAT"] = "5"; e["CLIENT_REPEAT_PRINT"] = "1"
    if os.environ.get("LAPS"): e["FLUHIP_CLIENT_TIMING"] = "1"
    args = [drv, "run", inp, frames, chans, 2048, 512, 2048, 32, 100, 42, 0, 0, int(os.environ.get("ASYNC", "0")), 0, -1, 0, -1, os.path.join(d, "o")]
    out = subprocess.run([str(x) for x in args], capture_output=True, text=True, env=e)
    print(env, [l for l in out.stdout.splitlines() if "elapsed" in l or "result" in l or "wall" in l], out.stderr[-900:])
