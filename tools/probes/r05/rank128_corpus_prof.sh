#!/bin/bash
# the bench corpus (128 x 10 s) at rank 128 kernel by kernel: rocprofv3 stats of 20 iterations
cd "$(dirname "$0")/../../.." || exit 1
export TMPDIR=/tmp; out=gpurun_out/r128; mkdir -p $out; d=$out/ks; rm -rf $d
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o ks -- python bench.py --rank 128 --iters 20 --steps 2 --warmup 1 --no-cpu-baseline --configs none > $out/ks.log 2>&1
find $d -name '*kernel_stats.csv' -exec cp {} $out/r128_kernel_stats.csv \;
rm -rf $d
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/r128/r128_kernel_stats.csv')):
    if float(r['TotalDurationNs']) > 5e5: print(f"  {r['Name'].split('(')[0][:64]:66s} {r['Calls']:>5} avg {float(r['AverageNs'])/1e3:8.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
