#!/bin/bash
# round 4: fused feature kernel with several frames per wavefront in one instruction stream (A/B build): timing + equality
export TMPDIR=/tmp; out=gpurun_out/r04p6; mkdir -p $out
for v in "FLUHIP_FEAT_FPW=1" "FLUHIP_FEAT_FPW=2" "FLUHIP_FEAT_FPW=22" "FLUHIP_FEAT_FPW=3" "FLUHIP_FEAT_FPW=1" "FLUHIP_FEAT_FPW=2"; do
  env FLUHIP_AB=1 $v python - <<'PY' | tee -a $out/c5_fpw.txt
import os, sys, json, subprocess
sys.path.insert(0, "flucoma-core_amd"); sys.path.insert(0, "oracle")
import numpy as np, fluhip, oracle_np
ctx = fluhip.Context(0)
x = np.stack([oracle_np.synth_audio(88200, 1000 + b) for b in range(8)])
got = ctx.bufmfcc(x, 1024, 1024, 512)
ref = oracle_np.bufmfcc_channel(x[3], 1024, 1024, 512)
err = float(np.abs(got[3] - ref).max() / np.abs(ref).max())
mb = ctx.bufmelbands(x[:, :20000], 1024, 1024, 512, n_bands=24, normalize=True, scale_db=False)
mref = oracle_np.bufmelbands_channel(x[5, :20000], 1024, 1024, 512, n_bands=24, normalize=True, scale_db=False)
merr = float(np.abs(mb[5] - mref).max() / np.abs(mref).max())
r = subprocess.run([sys.executable, "tools/bench_configs.py", "c5", "--no-cpu"], capture_output=True, text=True)
d = json.loads(r.stdout.splitlines()[-1])
print("c5 FPW=%s  %.3f ms  mfcc rel err %.2e  melbands rel err %.2e  checksum %.6f" % (os.environ.get("FLUHIP_FEAT_FPW"), d["ms"], err, merr, float(got.astype(np.float64).sum())))
PY
done
