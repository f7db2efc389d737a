#!/bin/bash
# round 4: the Nyquist bin as a side column of the strip schedule's combine step (A/B build: FLUHIP_STRIP_SIDE=0 is the old form)
export TMPDIR=/tmp; out=gpurun_out/r04p7; mkdir -p $out
python -m pytest tests/test_gpu_strip.py tests/test_gpu_configs.py tests/test_gpu_parity.py -q -k "strip or c2_full or c1_on or c1_shape or single_buffer_full" -x 2>&1 | tail -6
for v in "FLUHIP_STRIP_SIDE=1" "FLUHIP_STRIP_SIDE=0" "FLUHIP_STRIP_SIDE=1" "FLUHIP_STRIP_SIDE=0"; do
  env FLUHIP_AB=1 $v python tools/bench_configs.py c2 c1 --no-cpu 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['config'], '[$v]', round(d['us_per_iteration'],2), 'us/it', d['kernel_ms_per_iteration'])" | tee -a $out/strip_side.txt
done
