#!/bin/bash
# round 4: the iterations of the bench shard in ONE launch (nmf_iterate5_kernel: per-buffer barriers instead of kernel
# boundaries) against two launches per iteration (A/B build: FLUHIP_PERSIST=0); same box, alternating
export TMPDIR=/tmp; out=gpurun_out/r04p14; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_variants.py tests/test_gpu_parity.py -q -x -k "SIDE_NORM or NO_LAZY or corpus or c4" 2>&1 | tail -3
for v in 0 1 0 1 0 1; do
  env FLUHIP_AB=1 FLUHIP_PERSIST=$v timeout 300 python bench.py --no-cpu-baseline --configs none 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('persist=$v', round(d['value']), 'buffer-iterations/s', round(d['ms_per_step'],2), 'ms/step; launches', r['launches'], 'avg', round(r['avg_launch_ms']*1e3,1), 'us; frac', round(r['frac'],4), '; checksum', d.get('result_checksum'), d['result_finite'])" | tee -a $out/persist.txt
done
