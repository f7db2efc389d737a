#!/bin/bash
# config 3 with the side column's slices in the row-per-lane-group form (side_rows_kernel, default) against the row-per-thread
# kernel (FLUHIP_SIDE_ROWS=0, A/B build): us per iteration alternating, rocprofv3 duration of the launch, then the GPU tests
# that cross the side column at ranks 64 / 128
cd "$(dirname "$0")/../../.." || exit 1
export TMPDIR=/tmp FLUHIP_AB=1
out=gpurun_out/c3rows; mkdir -p $out
one() { python tools/bench_configs.py c3 --no-cpu 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print(round(j['us_per_iteration'],1), round(j['roofline']['frac'],4))"; }
for rep in 1 2 3; do
  for b in 0 1; do echo "c3 side_rows=$b: $(FLUHIP_SIDE_ROWS=$b one)"; done
done
d=$out/ks; rm -rf $d
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o ks -- python tools/bench_configs.py c3 --no-cpu > $out/ks.log 2>&1
find $d -name '*kernel_stats.csv' -exec cp {} $out/c3_kernel_stats.csv \;
rm -rf $d
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/c3rows/c3_kernel_stats.csv')):
    if float(r['TotalDurationNs']) > 1e6: print(f"  {r['Name'].split('(')[0][:64]:66s} {r['Calls']:>5} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
unset FLUHIP_AB
python -m pytest tests -x -q -m gpu -k "${TESTS:-c3 or wide or side or variants or long_factors or random or rank or offsize}" 2>&1 | tail -3
