#!/bin/bash
# where finalize<1>'s 24 us go on config 3: rocprofv3 kernel stats with the statistics section switched off (FLUHIP_FIN_DBG=1, A/B
# build: wrong norms, timing only) against the normal kernel
cd "$(dirname "$0")/../../.." || exit 1
export TMPDIR=/tmp FLUHIP_AB=1
out=gpurun_out/c3findbg; mkdir -p $out
for dbg in ${DBGS:-0 2 4 8 14}; do
  d=$out/ks$dbg; rm -rf $d
  FLUHIP_FIN_DBG=$dbg timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o ks -- python tools/bench_configs.py c3 --no-cpu > $out/ks$dbg.log 2>&1
  find $d -name '*kernel_stats.csv' -exec cp {} $out/c3_dbg${dbg}_kernel_stats.csv \;
  rm -rf $d
  echo "== FLUHIP_FIN_DBG=$dbg"
  python - $out/c3_dbg${dbg}_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'finalize' in r['Name'] or 'prereduce' in r['Name'] or 'side_slices' in r['Name']: print(f"  {r['Name'].split('(')[0][:64]:66s} {r['Calls']:>5} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:7.1f}")
PY
done
