#!/bin/bash
# round 4, review item 6: side column + norm combine as ONE launch, one workgroup per buffer (side_norm_kernel)
# (A/B build: FLUHIP_SIDE_NORM=0 the two launches as before); same box, alternating
export TMPDIR=/tmp; out=gpurun_out/r04p9; mkdir -p $out
python -m pytest tests/test_gpu_variants.py -q -x -k "SIDE_NORM or SIDE_STREAM or NO_LAZY or SIDE_SLICES" 2>&1 | tail -3
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -x -k "corpus or c4 or rehears or pool" 2>&1 | tail -3
for v in 0 1 0 1 0 1; do
  env FLUHIP_AB=1 FLUHIP_SIDE_NORM=$v python bench.py --no-cpu-baseline --configs none 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('side_norm=$v', round(d['value']), 'buffer-iterations/s', round(d['ms_per_step'],2), 'ms/step; update launch', round(r['avg_launch_ms']*1e3,1), 'us; between', round(d['schedule']['between_updates_ms_per_iteration']*1e3,2), 'us; checksum', d.get('result_checksum'))" | tee -a $out/side_norm.txt
done
