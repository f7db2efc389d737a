#!/bin/bash
# per-kernel average duration of the factor-update launches for a set of env settings
# usage: tools/kstats.sh "ENV=1 OTHER=2" "ENV=0" ...   (set KSTATS_ALL=1 to list every kernel)
export TMPDIR=/tmp
for cfg in "$@"; do
  rm -rf gpurun_out/ks; mkdir -p gpurun_out/ks
  env $cfg timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks -o b -- python bench.py --steps 1 --warmup 1 --iters 50 --no-cpu-baseline > gpurun_out/ks/log 2>&1
  echo "== $cfg"
  python - <<'PY'
import csv, os
allk = os.environ.get("KSTATS_ALL")
tot = 0.0
for r in csv.DictReader(open("gpurun_out/ks/b_kernel_stats.csv")):
    n = r["Name"]
    if any(t in n for t in ("nmf_update", "colstats", "colscale")): tot += float(r["TotalDurationNs"]) / 100.0 / 1e3
    if allk or any(t in n for t in ("nmf_update", "colstats", "colscale", "stft")):
        print("   %-64s calls %5s avg_us %8.1f" % (n[:64], r["Calls"], float(r["AverageNs"]) / 1e3))
print("   per-iteration kernel time (update + normalise): %.1f us" % tot)
PY
done
