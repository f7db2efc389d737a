#!/bin/bash
# per-kernel times (rocprofv3 --kernel-trace --stats) of the ragged corpus of tools/ragged_timing.py under schedule overrides:
#   tools/ragged_sweep.sh <outdir> "<ragged_timing args>" "ENV=.. ENV=.." ["ENV=.." ...]
out=$1; shift; args=$1; shift
export TMPDIR=/tmp
mkdir -p "$out"
i=0
for envs in "$@"; do
  d="$out/s$i"; rm -rf "$d"
  env $envs rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o ks -- python tools/ragged_timing.py $args > "$out/s$i.json" 2> "$out/s$i.err"
  echo "== [$envs] $(python -c "import json,sys; r=json.load(open('$out/s$i.json')); k=[x for x in ('ragged','equal') if x in r][0]; print(k, round(r[k]['us_per_iteration'],1), r[k]['plan'])")"
  python - "$d/ks_kernel_stats.csv" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name'].replace('fluhip::','').replace('void ','')[:44]
    if int(r['Calls'])>=100: print(f"   {n:46s} calls {r['Calls']:>5} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
  rm -rf "$d"
  i=$((i+1))
done
