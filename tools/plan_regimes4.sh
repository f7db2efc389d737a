export FLUHIP_AB=1   # the build whose experiment switches are live (flucoma-core_amd/build.py --ab)
# rank 128, few buffers: uniform schedule (FLUHIP_LIST_PLAN=0) against work lists (=1)
while read B secs K it; do
  for v in "FLUHIP_LIST_PLAN=0" "FLUHIP_LIST_PLAN=1"; do
    echo "B=$B secs=$secs K=$K $v: $(env $v timeout 300 python tools/batch_timing.py $B $secs $K $it 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); p = d['plan']; print(round(d['us_per_iteration'],1), 'us/it, with progress', round(d['us_per_iteration_progress'],1), 'splits', p['split_w'], p['split_h'], 'tail', p['tail_h'], 'strips_w', p['strips_w'])")"
  done
done <<'LIST'
3 10 128 30
4 10 128 30
8 10 128 30
12 10 128 30
4 60 128 20
1 60 128 30
LIST
