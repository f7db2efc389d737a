#!/bin/bash
# round 4: config 2 with the W update as a bin-strip launch against the fused form (A/B build), and the strip tests
export TMPDIR=/tmp; out=gpurun_out/r04p4; mkdir -p $out
python -m pytest tests/test_gpu_strip.py tests/test_gpu_configs.py -q -k "strip or c2_full" -x 2>&1 | tail -8
for v in "FLUHIP_STRIP_BIN=1" "FLUHIP_STRIP_BIN=0" "FLUHIP_STRIP_BIN=1" "FLUHIP_STRIP_BIN=0"; do
  env FLUHIP_AB=1 $v python tools/bench_configs.py c2 c1 --no-cpu 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['config'], '[$v]', round(d['us_per_iteration'],2), 'us/it', d['kernel_ms_per_iteration'], d['schedule']['strip'])" | tee -a $out/c2_bin.txt
done
