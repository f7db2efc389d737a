#!/bin/bash
# fft 2048 STFT of the bench corpus with SIXTEEN wavefronts per workgroup (FLUHIP_STFT_NW=16, A/B build: four per SIMD at 128
# registers -- 41 of them spilled --, sixteen frames per block = full 128-byte lines of the bin-major copy, one set of staging
# buffers, no sample prefetch; the run on file also had twelve wavefronts, since removed) against the production form (8 wavefronts, two sets), alternating
cd "$(dirname "$0")/../../.." || exit 1
export FLUHIP_AB=1
for rep in 1 2 3; do
  for nw in 8 16; do
    r=$(FLUHIP_STFT_NW=$nw python tools/stft_timing.py 2048 128 10 512 2>&1 | tail -1)
    echo "nw=$nw: $r"
  done
done
