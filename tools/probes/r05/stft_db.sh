#!/bin/bash
# fft 2048 STFT of the bench corpus with two sets of staging buffers used in turn (FLUHIP_STFT_DB=1, A/B build: one workgroup
# barrier per round instead of two, the bin-major flush of round r under the transforms of round r + 1; the window through the
# L1 to make room) against the production form, alternating; the checksum shows identical results
cd "$(dirname "$0")/../../.." || exit 1
export FLUHIP_AB=1
for rep in 1 2 3; do
  for db in ${DBS:-0 1}; do
    r=$(FLUHIP_STFT_DB=$db python bench.py --iters 1 --steps 20 --warmup 2 --no-cpu-baseline --configs none 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print(round(j['roofline_stft']['avg_launch_ms']*1e3,1), j['result_checksum'])")
    echo "db=$db stft us per launch, checksum: $r"
  done
done
