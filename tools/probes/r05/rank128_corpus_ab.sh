#!/bin/bash
# the bench corpus at rank 128 (round-major windows of 16 buffers): the pre-reduction in front of the norm combine taking the
# column sums of W' for the H update (default) against the H update's own pre-pass (FLUHIP_WNORM_PRE=2, A/B build), ms per 50
# iterations alternating; then 64 x 10 s and the stereo / single shapes for reference
cd "$(dirname "$0")/../../.." || exit 1
export FLUHIP_AB=1
run() { python bench.py --buffers $1 --rank $2 --iters 50 --steps 2 --warmup 1 --no-cpu-baseline --configs none 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print(round(j['ms_per_step'],2), j['result_checksum'])"; }
for rep in 1 2 3; do
  for pre in 2 1; do echo "128 x 10 s rank 128 wnorm_pre=$pre: $(FLUHIP_WNORM_PRE=$pre run 128 128)"; done
done
for shape in "64 128" "16 128" "128 112"; do
  set -- $shape
  for pre in 2 1; do echo "$1 x 10 s rank $2 wnorm_pre=$pre: $(FLUHIP_WNORM_PRE=$pre run $1 $2)"; done
done
