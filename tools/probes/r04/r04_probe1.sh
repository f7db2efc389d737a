#!/bin/bash
# round 4, first probes: the fused feature kernel at 16 / 12 / 8 wavefronts per workgroup (config 5), and config 2's strip
# kernel in its three launch forms (kernel trace).  Output under gpurun_out/r04p1/.
export TMPDIR=/tmp; out=gpurun_out/r04p1; mkdir -p $out
for nw in 16 12 8; do
  FLUHIP_AB=1 FLUHIP_FEAT_NW=$nw python tools/bench_configs.py c5 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('c5 NW=$nw', round(d['ms'],3), 'ms', d['kernel_ms'])" | tee -a $out/c5_nw.txt
done
d=$out/c2trace; rm -rf $d
rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python tools/c2_launch_forms.py > $out/c2trace.log 2>&1
python - <<'PY' | tee $out/c2_launch_forms.txt
import csv, glob
f = glob.glob('gpurun_out/r04p1/c2trace/**/*kernel_trace.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
seq = [(r['Kernel_Name'].split('(')[0][-60:], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, int(r['Start_Timestamp'])) for r in rows]
# the last five calls of three iterations: 7 launches each
tail = [s for s in seq if 'strip' in s[0]][-35:]
for i, (k, us, st) in enumerate(tail):
    gap = (st - tail[i-1][2]) / 1e3 - tail[i-1][1] if i else 0.0
    print(f'{i % 7}  {k:60s} {us:8.2f} us   gap before {gap:7.2f} us')
PY
