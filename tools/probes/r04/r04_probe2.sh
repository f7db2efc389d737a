#!/bin/bash
# round 4: config 5's fused feature kernel under two A/B switches (window pairs through the L1; per-wavefront issue priorities)
export TMPDIR=/tmp; out=gpurun_out/r04p2; mkdir -p $out
for v in "X=0" "FLUHIP_FEAT_WINDOW_GLOBAL=1" "FLUHIP_FEAT_PRIO=1" "FLUHIP_FEAT_WINDOW_GLOBAL=1 FLUHIP_FEAT_PRIO=1" "X=0" "FLUHIP_FEAT_PRIO=1"; do
  env FLUHIP_AB=1 $v python tools/bench_configs.py c5 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('c5 [$v]', round(d['ms'],3), 'ms', d['kernel_ms'])" | tee -a $out/c5_ab.txt
done
