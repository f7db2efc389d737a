#!/bin/bash
# alternating A/B of library builds on ONE box: tools/ab_libs.sh <outdir> <rows> <reps> <rounds> lib1 lib2 ...
# every library runs tools/perf_matrix.py on the same rows, in turn, `rounds` times (FLUHIP_LIB selects the build)
out=$1; rows=$2; reps=$3; rounds=$4; shift 4
mkdir -p $out
for r in $(seq 1 $rounds); do
  for lib in "$@"; do
    tag=$(basename $lib .so | sed 's/libflucoma_hip_\?//'); [ -z "$tag" ] && tag=prod
    FLUHIP_LIB=$lib python tools/perf_matrix.py --rows $rows --reps $reps --out $out/${tag}_r$r.json > /dev/null 2> $out/${tag}_r$r.err
  done
done
python - "$out" <<'PY'
import json, glob, os, sys
out = sys.argv[1]
tab = {}
for f in sorted(glob.glob(os.path.join(out, "*_r*.json"))):
    tag = os.path.basename(f)[:-5]
    for n, r in json.load(open(f))["rows"].items():
        v = r.get("us", r.get("ms"))
        tab.setdefault(n, {})[tag] = (round(v["min"], 3) if isinstance(v, dict) else r.get("error"), r.get("sustained_mhz") and round(r["sustained_mhz"]))
for n, d in tab.items():
    print(n)
    for t, v in sorted(d.items()):
        print("   %-16s %s" % (t, v))
PY
