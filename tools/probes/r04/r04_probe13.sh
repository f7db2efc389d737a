#!/bin/bash
# the profiled step's event pairs: does an (empty) event pair between the W and the H update lengthen what the H update's own
# event pair measures?  lib_prev/ = with the empty pair, lib/ = without.  Alternating, one box.
export TMPDIR=/tmp; out=gpurun_out/r04p13; mkdir -p $out
for v in lib_prev lib lib_prev lib; do
  env FLUHIP_LIB=$PWD/flucoma-core_amd/$v/libflucoma_hip.so python bench.py --no-cpu-baseline --configs none 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v', round(d['value']), round(d['ms_per_step'],2), 'ms/step; events', round(r['avg_launch_ms']*1e3,2), 'us per launch, frac', round(r['frac'],4), '; stamps W/H', round(r['clock_stamps']['w']['cycles_per_launch']/r['sustained_mhz'],2), round(r['clock_stamps']['h']['cycles_per_launch']/r['sustained_mhz'],2), 'us; profiled step', round(r['profiled_step_ms'],2), d['schedule'].get('between_updates_ms_per_iteration'))" | tee -a $out/events.txt
done
