#!/bin/bash
# rank 56 (M = 14 on the arrays of rank 64): the two-operand-set pipeline form (a hand-built lib_ab/libflucoma_hip_m14.so) against
# the in-place form the first off-size build used, bench corpus, 50 iterations per step, alternating; the checksum shows identical results
cd "$(dirname "$0")/../../.." || exit 1
for rep in 1 2 3; do
  for lib in ab m14; do
    r=$(FLUHIP_LIB=flucoma-core_amd/lib_ab/libflucoma_hip_$lib.so python bench.py --rank 56 --iters 50 --steps 3 --warmup 1 --no-cpu-baseline --configs none 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print(round(j['ms_per_step'],2), round(j['roofline']['avg_launch_ms']*1e3,1), j['result_checksum'])")
    echo "rank 56 lib=$lib: $r"
  done
done
