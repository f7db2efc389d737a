#!/bin/bash
export FLUHIP_AB=1   # the build whose experiment switches are live (flucoma-core_amd/build.py --ab)
# tools/batch_plan_sweep.sh <B> <plans...>  with plan = W:SPLIT (strips per buffer : contraction splits; 0 = planner's own)
B=$1; shift
for plan in "$@"; do
  w=${plan%%:*}; sp=${plan##*:}
  echo "W=$w SPLIT=$sp $(env FLUHIP_PLAN_W=$w FLUHIP_PLAN_SPLIT=$sp python tools/batch_timing.py $B 10 32 100 | python -c "import json,sys; r=json.loads(sys.stdin.readline()); print(round(r['us_per_iteration'],1), r['plan']['split_w'], r['plan']['split_h'], r['plan']['strips_w'])")"
done
