#!/bin/bash
# round 4: the strip schedule's kernels compiled with -mllvm -amdgpu-mfma-vgpr-form (MFMA results in VGPRs: no v_accvgpr_read
# in front of every VALU use) against the default (AGPR destinations); lib_prev/ = before, lib/ = with the flag; one box
export TMPDIR=/tmp; out=gpurun_out/r04p15; mkdir -p $out
python -m pytest tests/test_gpu_strip.py tests/test_gpu_configs.py tests/test_gpu_parity.py -q -x -k "strip or c2_full or c1_on or c1_shape or single_buffer_full" 2>&1 | tail -3
for v in lib_prev lib lib_prev lib lib_prev lib; do
  env FLUHIP_LIB=$PWD/flucoma-core_amd/$v/libflucoma_hip.so python tools/bench_configs.py c2 c1 --no-cpu 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['config'], '[$v]', round(d['us_per_iteration'],2), 'us/it', d['kernel_ms_per_iteration'])" | tee -a $out/strip_vgpr_form.txt
done
