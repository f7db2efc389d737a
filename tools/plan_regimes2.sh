export FLUHIP_AB=1   # the build whose experiment switches are live (flucoma-core_amd/build.py --ab)
# one and two long buffers at rank 32, and 16 - 40 buffers at rank 128: uniform schedule (FLUHIP_LIST_PLAN=0) against work lists (=1)
while read B secs K it; do
  for v in "FLUHIP_LIST_PLAN=0" "FLUHIP_LIST_PLAN=1"; do
    echo "B=$B secs=$secs K=$K $v: $(env $v timeout 300 python tools/batch_timing.py $B $secs $K $it 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); p = d['plan']; print(round(d['us_per_iteration'],1), 'us/it, with progress', round(d['us_per_iteration_progress'],1), 'splits', p['split_w'], p['split_h'], 'tail', p['tail_h'], 'strips_w', p['strips_w'])")"
  done
done <<'LIST'
1 30 32 60
1 60 32 60
1 120 32 60
1 300 32 40
2 30 32 60
2 60 32 60
2 120 32 40
16 10 128 20
24 10 128 20
32 10 128 20
LIST
