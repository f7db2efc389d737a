#!/bin/bash
# round 4, config 5: two workgroups of ten (nine) wavefronts per CU -- five wavefronts per SIMD -- against one of sixteen
export TMPDIR=/tmp; out=gpurun_out/r04p11; mkdir -p $out
FLUHIP_FEAT_NW=10 python -m pytest tests/test_gpu_configs.py -q -x -k "c5" 2>&1 | tail -2
for v in 16 10 9 16 10 9; do
  env FLUHIP_AB=1 FLUHIP_FEAT_NW=$v python tools/bench_configs.py c5 --no-cpu 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('c5 [FLUHIP_FEAT_NW=$v]', round(d['ms'],3), 'ms', d.get('checksum'))" | tee -a $out/c5_nw.txt
done
