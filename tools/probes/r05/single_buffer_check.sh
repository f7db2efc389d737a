#!/bin/bash
# single 10 s buffers (BASELINE config 4's buffer alone) at ranks 32, 64, 128: ms per 200 iterations, this library against the
# library of the commit at the start of the session (lib_ab/libflucoma_hip_prev.so), alternating
cd "$(dirname "$0")/../../.." || exit 1
for rep in 1 2; do
  for k in 32 64 128; do
    for lib in prev new; do
      L=""; [ $lib = prev ] && L=flucoma-core_amd/lib_ab/libflucoma_hip_prev.so
      echo "1 x 10 s rank $k $lib: $(FLUHIP_LIB=$L python bench.py --buffers 1 --rank $k --iters 200 --steps 2 --warmup 1 --no-cpu-baseline --configs none 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print(round(j['ms_per_step'],3), 'ms', j['result_checksum'])")"
    done
  done
done
