"""wall time of a mono 10 s rank-32 BufNMF job through the C++ client, without and with resynthesis: python tools/client_mono_timing.py"""
import os, sys, subprocess, tempfile
os.environ.setdefault("FLUHIP_AB", "1")   # the build whose experiment switches are live (build.py --ab)
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flucoma-core_amd"))
import synth
frames, chans = 441000, 1
audio = synth.synth_audio(frames, 500).astype(np.float32)
d = tempfile.mkdtemp(); inp = os.path.join(d, "in.f32"); audio.tofile(inp)
drv = os.path.join(ROOT, "flucoma-core_amd", "lib", "client_driver")
for env in ({}, {"CLIENT_RESYNTH": "1"}, {"CLIENT_RESYNTH": "1", "FLUHIP_PINNED_D2H": "0"}):
    e = dict(os.environ); e.update(env); e["CLIENT_REPEAT"] = "5"; e["CLIENT_REPEAT_PRINT"] = "1"
    args = [drv, "run", inp, frames, chans, 2048, 512, 2048, 32, 100, 42, 0, 0, 0, 0, -1, 0, -1, os.path.join(d, "o")]
    out = subprocess.run([str(x) for x in args], capture_output=True, text=True, env=e)
    print(env, [l for l in out.stderr.splitlines() if "repeat" in l][-2:])
