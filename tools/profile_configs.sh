#!/bin/bash
# rocprofv3 summaries for the BASELINE configs other than the bench shard (c2, c3, c5, c1, c4x1):
#   tools/profile_configs.sh <outdir> <config> [<config> ...]
# per config: <outdir>/<cfg>.json (the tools/bench_configs.py line, un-profiled), <cfg>_kernel_stats.csv
# (rocprofv3 --kernel-trace --stats of the same command) and <cfg>_pmc.json (separate --pmc passes: FETCH_SIZE,
# WRITE_SIZE + TCC hit/miss, SQ busy/wait -- never combined with tracing), summarised per kernel.
out=$1; shift
export TMPDIR=/tmp
mkdir -p "$out"
for cfg in "$@"; do
  python tools/bench_configs.py "$cfg" > "$out/$cfg.json" 2> "$out/$cfg.err" || { echo "$cfg: bench failed"; tail -5 "$out/$cfg.err"; continue; }
  d="$out/_$cfg"; rm -rf "$d"; mkdir -p "$d"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o ks -- python tools/bench_configs.py "$cfg" --no-cpu > "$d/ks.log" 2>&1
  cp "$d"/ks_kernel_stats.csv "$out/${cfg}_kernel_stats.csv" 2>/dev/null || find "$d" -name '*kernel_stats.csv' -exec cp {} "$out/${cfg}_kernel_stats.csv" \;
  pass() { name=$1; shift; timeout 600 rocprofv3 --pmc "$@" --output-format csv -d "$d" -o "$name" -- python tools/bench_configs.py "$cfg" --no-cpu > "$d/$name.log" 2>&1; }
  pass fetch FETCH_SIZE
  pass tcc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
  pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
  pass sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
  python - "$d" "$out/${cfg}_pmc.json" "$cfg" <<'PY'
import collections, csv, glob, json, sys
d, outp, cfg = sys.argv[1:4]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0][:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {"source": "rocprofv3 --pmc, one counter group per pass, no tracing combined (tools/profile_configs.sh); "
                 "per-launch averages per kernel; FETCH_SIZE / WRITE_SIZE in KB as reported; hbm_bytes = (2 x FETCH_SIZE + "
                 "WRITE_SIZE) x 1024 -- the gfx950 correction of MI355X_MICROARCH.md's HBM section for 16-byte-per-lane "
                 "streaming loads, cross-checked by TCC_MISS_sum x 128 B",
       "config": cfg, "kernels": {}}
for k, cs in agg.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    row = {"launches": max(len(v) for v in cs.values()), **{c: round(x, 2) for c, x in m.items()}}
    if "FETCH_SIZE" in m:
        row["hbm_bytes_per_launch"] = (2 * m["FETCH_SIZE"] + m.get("WRITE_SIZE", 0)) * 1024
    if "TCC_MISS_sum" in m:
        row["tcc_miss_bytes_per_launch"] = m["TCC_MISS_sum"] * 128
    if m.get("SQ_VALU_MFMA_BUSY_CYCLES") and m.get("SQ_WAVE_CYCLES"):
        row["mfma_busy_fraction"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * m["SQ_WAVE_CYCLES"])
    res["kernels"][k] = row
json.dump(res, open(outp, "w"), indent=1)
PY
  rm -rf "$d"
  echo "== $cfg"; cat "$out/$cfg.json" | python -c "import json,sys; r=json.loads(sys.stdin.readline()); print({k: r.get(k) for k in ('value','us_per_iteration','stft_frames_per_s','ms','kernel_ms','kernel_ms_per_iteration')}, r.get('roofline'))"
done
