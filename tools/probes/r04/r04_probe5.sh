#!/bin/bash
export TMPDIR=/tmp; out=gpurun_out/r04p5; mkdir -p $out; d=$out/trace; rm -rf $d
FLUHIP_AB=1 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python tools/c2_launch_forms.py > $out/log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r04p5/trace/**/*kernel_stats.csv', recursive=True)
for r in csv.DictReader(open(f[0])):
    print(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3)
PY
