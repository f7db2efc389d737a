#!/bin/bash
# config 5 after the instruction trims of round 5, session 4 (one Goldschmidt step for the magnitudes, DPP moves without an
# `old` operand, the DCT quarter row unrolled for 37 .. 40 bands) against the library of the commit before
# (lib_ab/libflucoma_hip_prev.so: `git worktree add /tmp/old HEAD~; python flucoma-core_amd/build.py`), alternating on one
# box; then the full square root alone (FLUHIP_FEAT_FASTMAG=0, A/B build); then the feature tests on the new library
cd "$(dirname "$0")/../../.." || exit 1
one() { python tools/bench_configs.py c5 --no-cpu 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print(round(j['ms'],3), round(j['kernel_ms']['features'],3))"; }
for rep in 1 2 3; do
  echo "c5 prev: $(FLUHIP_LIB=flucoma-core_amd/lib_ab/libflucoma_hip_prev.so one)"
  echo "c5 new : $(one)"
done
for rep in 1 2; do
  echo "c5 ab fastmag=1: $(FLUHIP_AB=1 one)"
  echo "c5 ab fastmag=0: $(FLUHIP_AB=1 FLUHIP_FEAT_FASTMAG=0 one)"
done
python -m pytest tests -q -m gpu -k "mfcc or melbands or stft or feature or c5" 2>&1 | tail -3
