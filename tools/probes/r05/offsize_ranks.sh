#!/bin/bash
# off-size ranks: the bench shard's corpus (128 x 10 s, fft 2048 / hop 512, 50 iterations) at ranks between the array ranks with
# the off-size forms (default) against the padded forms of rank 32 / 64 / 128 (FLUHIP_OFFSIZE=0), A/B build, one box, alternating
#     bash tools/probes/r05/offsize_ranks.sh [ranks ...]        (default: one rank per form and the padded ranks beside them)
cd "$(dirname "$0")/../../.." || exit 1
export FLUHIP_AB=1
RANKS=${*:-"20 32 40 48 56 64 80 96 100 128"}
for rep in 1 2; do
  for K in $RANKS; do
    for off in 1 0; do
      r=$(FLUHIP_OFFSIZE=$off python bench.py --rank $K --iters 50 --steps 3 --warmup 1 --no-cpu-baseline --configs none 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print(round(j['ms_per_step'],2), round(j['roofline']['avg_launch_ms']*1e3,1))")
      echo "rank $K offsize=$off: ms per step (50 it), us per update launch: $r"
    done
  done
done
