#!/bin/bash
# config 3 with the side-column launch in front of the W update and the update's column sums of H taken from its slices'
# denominators (default) against the pre-pass over H (FLUHIP_COLSUM_FROM_SIDE=0, A/B build): us per iteration alternating,
# rocprofv3 kernel stats of the default, then the GPU tests that cross rank-128 corpora
cd "$(dirname "$0")/../../.." || exit 1
export TMPDIR=/tmp FLUHIP_AB=1
out=gpurun_out/c3sf; mkdir -p $out
one() { python tools/bench_configs.py c3 --no-cpu 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print(round(j['us_per_iteration'],1), {k: round(v,4) for k,v in j['kernel_ms_per_iteration'].items()})"; }
for rep in 1 2 3; do
  for b in 0 1; do echo "c3 colsum_from_side=$b: $(FLUHIP_COLSUM_FROM_SIDE=$b one)"; done
done
d=$out/ks; rm -rf $d
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o ks -- python tools/bench_configs.py c3 --no-cpu > $out/ks.log 2>&1
find $d -name '*kernel_stats.csv' -exec cp {} $out/c3_kernel_stats.csv \;
rm -rf $d
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/c3sf/c3_kernel_stats.csv')):
    if float(r['TotalDurationNs']) > 1e6: print(f"  {r['Name'].split('(')[0][:64]:66s} {r['Calls']:>5} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
unset FLUHIP_AB
python -m pytest tests -x -q -m gpu -k "c3 or wide or split or list or ragged or variants or side or offsize or rank" 2>&1 | tail -3
