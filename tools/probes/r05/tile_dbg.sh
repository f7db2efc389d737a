#!/bin/bash
# timing experiments of the bin-tiled W update (A/B build, FLUHIP_TILE_DBG: results are wrong, durations are what is read):
# rocprofv3 kernel durations of config 2 with the H refills, the V refills or both switched off
cd "$(dirname "$0")/../../.." || exit 1
export TMPDIR=/tmp FLUHIP_AB=1
out=gpurun_out/tile_dbg; mkdir -p $out
for d in ${*:-0 1 2 3}; do
  rm -rf $out/_d$d
  FLUHIP_TILE_DBG=$d timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/_d$d -o ks -- python tools/bench_configs.py c2 --no-cpu > $out/d$d.log 2>&1
  f=$(find $out/_d$d -name '*kernel_stats.csv' | head -1)
  python - "$f" "$d" <<'PY'
import csv, sys
rows = {r["Name"].split("(")[0][-60:]: float(r["AverageNs"]) for r in csv.DictReader(open(sys.argv[1]))}
print("dbg", sys.argv[2], {k: round(v / 1e3, 2) for k, v in rows.items() if "bintile_kernel<" in k or "nmf_strip_kernel" in k})
PY
  rm -rf $out/_d$d
done
