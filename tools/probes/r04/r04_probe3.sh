#!/bin/bash
# round 4: config 2 on every schedule family the planner has (A/B build): frame strips (default), split contraction, work lists
export TMPDIR=/tmp; out=gpurun_out/r04p3; mkdir -p $out
for v in "X=0" "FLUHIP_STRIP=0" "FLUHIP_STRIP=0 FLUHIP_LIST_PLAN=1" "FLUHIP_STRIP=0 FLUHIP_LIST_PLAN=1 FLUHIP_RG_GMAX=4"; do
  env FLUHIP_AB=1 $v python tools/bench_configs.py c2 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('c2 [$v]', round(d['us_per_iteration'],2), 'us/it', d['kernel_ms_per_iteration'], d['schedule'])" | tee -a $out/c2_families.txt
done
