export TMPDIR=/tmp; out=gpurun_out/c4x1; mkdir -p $out
python tools/bench_configs.py c4x1 --no-cpu > $out/c4x1.json 2>/dev/null
d=$out/ks; rm -rf $d
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o ks -- python tools/bench_configs.py c4x1 --no-cpu > $out/ks.log 2>&1
find $d -name '*kernel_stats.csv' -exec cp {} $out/c4x1_kernel_stats.csv \;
rm -rf $d
python - <<'PY'
import csv, json
j=json.loads(open('gpurun_out/c4x1/c4x1.json').readline()); print(j.get('us_per_iteration'), j.get('schedule'), j.get('kernel_ms_per_iteration'))
for r in csv.DictReader(open('gpurun_out/c4x1/c4x1_kernel_stats.csv')):
    if float(r['TotalDurationNs']) > 2e5: print(f"  {r['Name'].split('(')[0][:70]:72s} {r['Calls']:>5} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
