export FLUHIP_AB=1   # the build whose experiment switches are live (flucoma-core_amd/build.py --ab)
# the planner's choice against the forced forms over corpus shapes off the BASELINE configs: B buffers x seconds at rank K
# (tools/batch_timing.py; us per iteration of the whole corpus, without / with a progress callback)
while read B secs K it; do
  for v in "A=1" "FLUHIP_LIST_PLAN=0" "FLUHIP_LIST_PLAN=1"; do
    echo "B=$B secs=$secs K=$K $v: $(env $v timeout 300 python tools/batch_timing.py $B $secs $K $it 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); p = d['plan']; print(round(d['us_per_iteration'],1), 'us/it, with progress', round(d['us_per_iteration_progress'],1), 'splits', p['split_w'], p['split_h'], 'tail', p['tail_h'], 'strips_w', p['strips_w'])")"
  done
done <<'LIST'
1024 2 32 40
512 1 32 40
256 5 16 40
300 10 16 40
300 10 8 40
64 30 32 40
4 120 32 40
2 300 32 40
16 60 64 30
40 10 128 20
LIST
