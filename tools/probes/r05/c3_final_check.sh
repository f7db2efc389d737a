#!/bin/bash
# after the column-sum shortcuts: config 3 alternating (A/B build), a single 10 s buffer at rank 128 and the 128-buffer corpus
# at rank 128 with and without them, then the whole GPU suite on the production library
cd "$(dirname "$0")/../../.." || exit 1
export TMPDIR=/tmp
one() { FLUHIP_AB=1 python tools/bench_configs.py c3 --no-cpu 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print(round(j['us_per_iteration'],1), {k: round(v,4) for k,v in j['kernel_ms_per_iteration'].items()})"; }
for rep in 1 2; do
  for b in 0 1; do echo "c3 colsum_from_side=$b: $(FLUHIP_COLSUM_FROM_SIDE=$b one)"; done
done
for b in 0 1; do
  echo "128 x 10 s rank 128, shortcuts=$b: $(FLUHIP_AB=1 FLUHIP_COLSUM_FROM_SIDE=$b python bench.py --rank 128 --iters 50 --steps 2 --warmup 1 --no-cpu-baseline --configs none 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print(round(j['ms_per_step'],2), 'ms per 50 iterations', j['result_checksum'])")"
  echo "1 x 10 s rank 128, shortcuts=$b: $(FLUHIP_AB=1 FLUHIP_COLSUM_FROM_SIDE=$b python bench.py --buffers 1 --rank 128 --iters 200 --steps 2 --warmup 1 --no-cpu-baseline --configs none 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print(round(j['ms_per_step'],3), 'ms per 200 iterations', j['result_checksum'])")"
done
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
