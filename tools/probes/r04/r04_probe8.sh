#!/bin/bash
# round 4, review item 6 (first half): the W update's side column on a second stream beside the update launch
# (A/B build: FLUHIP_SIDE_STREAM=0 one stream as before, =1 forked / joined); same box, alternating
export TMPDIR=/tmp; out=gpurun_out/r04p8; mkdir -p $out
for v in 0 1 0 1 0 1; do
  env FLUHIP_AB=1 FLUHIP_SIDE_STREAM=$v python bench.py --no-cpu-baseline --configs none 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('side_stream=$v', round(d['value']), 'buffer-iterations/s', round(d['ms_per_step'],2), 'ms/step; update launch', round(r['avg_launch_ms']*1e3,1), 'us; checksum', d.get('result_checksum'), d.get('schedule'))" | tee -a $out/side_stream.txt
done
python -m pytest tests/test_gpu_variants.py -q -x -k "SIDE_STREAM" 2>&1 | tail -3
for v in 0 1 0 1; do
  env FLUHIP_AB=1 FLUHIP_SIDE_STREAM=$v python tools/bench_configs.py c3 --no-cpu 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['config'], '[side_stream=$v]', round(d['us_per_iteration'],2), 'us/it', d['kernel_ms_per_iteration'])" | tee -a $out/side_stream.txt
done
