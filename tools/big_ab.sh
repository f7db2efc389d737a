export FLUHIP_AB=1   # the build whose experiment switches are live (flucoma-core_amd/build.py --ab)
# equal-length corpora beyond two rounds of wavefronts (B x 10 s, rank 32): the planner's choice against the forced forms
for B in 288 400 520 900 1000; do
  for v in "A=1" "FLUHIP_LIST_PLAN=0" "FLUHIP_LIST_PLAN=1"; do
    echo "B=$B $v: $(env $v timeout 300 python tools/batch_timing.py $B 10 32 40 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(round(d['us_per_iteration'],1), 'us/it, with progress', round(d['us_per_iteration_progress'],1), d['plan'])")"
  done
done
