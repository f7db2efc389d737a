#!/bin/bash
# config 5 with the window pairs through the L1 instead of the LDS (FLUHIP_FEAT_WINDOW_GLOBAL=1, A/B build; measured equal in
# round 4 when the VALU bounded the kernel alone) re-measured after the instruction trims, alternating
cd "$(dirname "$0")/../../.." || exit 1
export FLUHIP_AB=1
one() { python tools/bench_configs.py c5 --no-cpu 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print(round(j['ms'],3), round(j['kernel_ms']['features'],3))"; }
for rep in 1 2 3; do
  echo "c5 window in the LDS: $(one)"
  echo "c5 window through the L1: $(FLUHIP_FEAT_WINDOW_GLOBAL=1 one)"
done
