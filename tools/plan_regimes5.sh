export FLUHIP_AB=1   # the build whose experiment switches are live (flucoma-core_amd/build.py --ab)
# other transform sizes: uniform schedule (FLUHIP_LIST_PLAN=0) against the planner's choice (A=1) and forced lists (=1)
while read B secs K it fft hop; do
  for v in "A=1" "FLUHIP_LIST_PLAN=0" "FLUHIP_LIST_PLAN=1"; do
    echo "B=$B secs=$secs K=$K fft=$fft hop=$hop $v: $(env $v timeout 300 python tools/batch_timing.py $B $secs $K $it $fft $hop 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); p = d['plan']; print(round(d['us_per_iteration'],1), 'us/it, with progress', round(d['us_per_iteration_progress'],1), 'splits', p['split_w'], p['split_h'], 'tail', p['tail_h'], 'strips_w', p['strips_w'])")"
  done
done <<'LIST'
200 10 32 40 1024 256
64 10 32 40 4096 1024
1 120 32 60 1024 256
1 20 32 60 1024 256
300 5 16 40 1024 512
LIST
