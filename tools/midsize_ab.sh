export FLUHIP_AB=1   # the build whose experiment switches are live (flucoma-core_amd/build.py --ab)
# equal-length corpora between one and two rounds of the 1024 SIMDs (10 s buffers, rank 32): the planner's choice (work lists), the
# uniform schedule (FLUHIP_LIST_PLAN=0; with FLUHIP_TAIL_SPLIT=0 without the two-launch H update)
for B in 144 176 200 232 250; do
  for v in "A=1" "FLUHIP_LIST_PLAN=0" "FLUHIP_LIST_PLAN=0 FLUHIP_TAIL_SPLIT=0"; do
    echo "B=$B $v: $(env $v timeout 300 python tools/batch_timing.py $B 10 32 100 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(round(d['us_per_iteration'],1), 'us/it, with progress', round(d['us_per_iteration_progress'],1), d['plan'])")"
  done
done
