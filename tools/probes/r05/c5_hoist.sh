#!/bin/bash
# config 5 with the band stage's lane constants (boundary slots, the lane's quarter row of the DCT table) kept in registers
# across the frame loop, against the library of the commit before (lib_ab/libflucoma_hip_trim1.so), alternating; feature tests
cd "$(dirname "$0")/../../.." || exit 1
one() { python tools/bench_configs.py c5 --no-cpu 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print(round(j['ms'],3), round(j['kernel_ms']['features'],3))"; }
for rep in 1 2 3; do
  echo "c5 before: $(FLUHIP_LIB=flucoma-core_amd/lib_ab/libflucoma_hip_trim1.so one)"
  echo "c5 hoisted: $(one)"
done
python -m pytest tests -q -m gpu -k "mfcc or melbands or feature or c5" 2>&1 | tail -2
