#!/bin/bash
# round 4, config 5: the 8 x 8 x 8 transform's exchange buffer addressed through bit swizzles (no bank conflicts behind either
# exchange) + the band / DCT stages' LDS reads kept single (ds_read_b64 at 256 B/clk instead of ds_read2 at 128).
# lib_prev/ = the library of the commit before (built by hand: git stash; build.py; cp), lib/ = this tree.  Alternating, one box.
export TMPDIR=/tmp; out=gpurun_out/r04p12; mkdir -p $out
python -m pytest tests -m gpu -q -x -k "feat or mfcc or melbands or c5 or stft or c1_" 2>&1 | tail -3
for v in lib_prev lib lib_prev lib lib_prev lib; do
  env FLUHIP_LIB=$PWD/flucoma-core_amd/$v/libflucoma_hip.so python tools/bench_configs.py c5 c1 --no-cpu 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d.get('config','?'), '[$v]', round(d.get('ms', d.get('us_per_iteration',0)),3), d.get('kernel_ms', ''))" | tee -a $out/c5_swizzle.txt
done
