# the C++ client driver built with -fsanitize=address,undefined (flucoma-core_amd/lib/client_driver_asan), every client,
# synchronous and threaded: bash tools/asan_client_run.sh   (build line in tools/README.md)
cd ${GRAFT_REPO_ROOT:-.}
python - <<'PY'
import numpy as np, sys, os
sys.path.insert(0,"flucoma-core_amd")
import synth
a = np.stack([synth.synth_audio(60000, 500+c) for c in range(3)], axis=1).astype(np.float32)
a.tofile("/tmp/in3.f32")
PY
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:use_sigaltstack=0   # (the HIP runtime and ASan both want the worker threads' alternate signal stack)
D=flucoma-core_amd/lib/client_driver_asan
for asyncm in 0 1; do
  echo "== nmf 3 channels async=$asyncm"; CLIENT_RESYNTH=1 $D run /tmp/in3.f32 60000 3 2048 512 2048 5 10 42 0 0 $asyncm 100 50000 0 -1 /tmp/o 2>&1 | head -60
  echo "== nmf mono async=$asyncm"; CLIENT_RESYNTH=1 $D run /tmp/in3.f32 60000 3 2048 512 2048 5 10 42 0 0 $asyncm 0 -1 1 1 /tmp/o1 2>&1 | head -60
done
echo "== stft"; $D stft /tmp/in3.f32 60000 3 1024 256 1024 1 0 -1 1 /tmp/s 2>&1 | head -40
echo "== mfcc"; $D mfcc /tmp/in3.f32 60000 3 1024 256 1024 1 40 13 0 0 -1 0 -1 1 /tmp/m 2>&1 | head -40
echo "== seed"; $D seed /tmp/in3.f32 180000 1024 256 1024 1 8 0.7 0 42 1 /tmp/sd 2>&1 | head -40
