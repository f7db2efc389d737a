#!/bin/bash
# round 4: the persistent form's two barriers (FLUHIP_PERSIST_LIGHT=1: no agent-scope fences) against two launches per iteration;
# needs profiles/r04/persistent_form.patch applied (the form was measured and not adopted)
for v in "FLUHIP_PERSIST=0" "FLUHIP_PERSIST_LIGHT=1" "FLUHIP_PERSIST_LIGHT=0" "FLUHIP_PERSIST=0" "FLUHIP_PERSIST_LIGHT=1"; do
  env FLUHIP_AB=1 $v timeout 300 python bench.py --no-cpu-baseline --configs none 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=r['clock_stamps']
print('$v', round(d['value']), round(d['ms_per_step'],2), 'W/H us', round(c['w']['cycles_per_launch']/c['w']['sustained_mhz'],1), round(c['h']['cycles_per_launch']/c['h']['sustained_mhz'],1), 'MHz', round(c['w']['sustained_mhz']), 'checksum', d.get('result_checksum'))"
done
