#!/bin/bash
# per-kernel average duration of the factor-update launches for a set of env settings
export TMPDIR=/tmp
for cfg in "$@"; do
  rm -rf gpurun_out/ks; mkdir -p gpurun_out/ks
  env $cfg timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks -o b -- python bench.py --steps 1 --warmup 1 --iters 50 --no-cpu-baseline > gpurun_out/ks/log 2>&1
  echo "== $cfg"
  python - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/ks/b_kernel_stats.csv")):
    if "nmf_update" in r["Name"] or "colnorm" in r["Name"] or "stft" in r["Name"]:
        print("   %-64s calls %5s avg_us %8.1f" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
