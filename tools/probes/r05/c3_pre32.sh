#!/bin/bash
# (the run on file had 32 pre-reduction groups per buffer -- since back at 16 -- and)
# config 3 with the column sums from the side slices in four column blocks
# per buffer: rocprofv3 kernel stats and the line's us per iteration (compare profiles/r05/cfg_v5/c3_kernel_stats.csv)
cd "$(dirname "$0")/../../.." || exit 1
export TMPDIR=/tmp
out=gpurun_out/c3pre32; mkdir -p $out
for rep in 1 2 3; do python tools/bench_configs.py c3 --no-cpu 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print('c3', round(j['us_per_iteration'],1), round(j['roofline']['frac'],4))"; done
d=$out/ks; rm -rf $d
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o ks -- python tools/bench_configs.py c3 --no-cpu > $out/ks.log 2>&1
find $d -name '*kernel_stats.csv' -exec cp {} $out/c3_kernel_stats.csv \;
rm -rf $d
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/c3pre32/c3_kernel_stats.csv')):
    if float(r['TotalDurationNs']) > 1e6: print(f"  {r['Name'].split('(')[0][:64]:66s} {r['Calls']:>5} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
python -m pytest tests -x -q -m gpu -k "c3 or side or long_factors" 2>&1 | tail -2
