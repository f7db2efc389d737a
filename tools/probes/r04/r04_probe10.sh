#!/bin/bash
# round 4, review item 6: the W update's side column formed in the epilogue of the H update in front (SIDEQ instantiations of
# nmf_update5_kernel; A/B build: FLUHIP_SIDE_FROM_H=0 the side-column launch as before); same box, alternating
export TMPDIR=/tmp; out=gpurun_out/r04p10; mkdir -p $out
python -m pytest tests/test_gpu_variants.py -q -x -k "SIDE_ or NORM_IN or NO_LAZY or LIST_PLAN" 2>&1 | tail -3
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -x -k "corpus or c4 or rehears or pool or client or channels" 2>&1 | tail -3
for v in "FLUHIP_SIDE_FROM_H=0" "FLUHIP_NORM_IN_H=0" "FLUHIP_NORM_IN_H=1" "FLUHIP_SIDE_FROM_H=0" "FLUHIP_NORM_IN_H=0" "FLUHIP_NORM_IN_H=1"; do
  env FLUHIP_AB=1 $v python bench.py --no-cpu-baseline --configs none 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v', round(d['value']), 'buffer-iterations/s', round(d['ms_per_step'],2), 'ms/step; update launch', round(r['avg_launch_ms']*1e3,1), 'us; W / H cycles', r['clock_stamps']['w']['cycles_per_launch'], r['clock_stamps']['h']['cycles_per_launch'], '; between', round(d['schedule']['between_updates_ms_per_iteration']*1e3,2), 'us; checksum', d.get('result_checksum'))" | tee -a $out/side_from_h.txt
done
