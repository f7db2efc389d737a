#!/bin/bash
export FLUHIP_AB=1   # the build whose experiment switches are live (flucoma-core_amd/build.py --ab)
export TMPDIR=/tmp
cfg=$1; shift
for plan in "$@"; do
  w=${plan%%:*}; sp=${plan##*:}
  rm -rf gpurun_out/ps; mkdir -p gpurun_out/ps
  env FLUHIP_PLAN_W=$w FLUHIP_PLAN_SPLIT=$sp rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ps -o p -- python tools/bench_configs.py $cfg > gpurun_out/ps/log 2>&1
  echo "== W=$w SPLIT=$sp  $(grep "^$cfg" gpurun_out/ps/log | sed 's/.*nmf/nmf/' | cut -c1-60)"
  python - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/ps/p_kernel_stats.csv")):
    if "nmf_update" in r["Name"]:
        print("     %-60s calls %5s avg_us %8.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
