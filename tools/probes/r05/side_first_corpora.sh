#!/bin/bash
# corpora at rank 128 (and 112: the other pre-pass form) whose side column and norm combine are one launch (side_norm_kernel):
# that launch against the side-first order (side slices in front of the W update, their denominators as its column sums of H;
# FLUHIP_SIDE_FIRST_CORPORA=1, A/B build), ms per 50 iterations, alternating
cd "$(dirname "$0")/../../.." || exit 1
export FLUHIP_AB=1
run() { python bench.py --buffers $1 --rank $2 --iters 50 --steps 2 --warmup 1 --no-cpu-baseline --configs none $3 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print(round(j['ms_per_step'],2), j['result_checksum'])"; }
for rep in 1 2; do
  for shape in "128 128" "64 128" "256 128" "128 112"; do
    set -- $shape
    for sf in 0 1; do echo "$1 x 10 s rank $2 side_first=$sf: $(FLUHIP_SIDE_FIRST_CORPORA=$sf run $1 $2)"; done
  done
done
