#!/bin/bash
# config 5 (fused feature kernel, fft 1024) and the fft-1024 STFT with the real-FFT split's partner bins fetched through the
# staging buffer (default, FLUHIP_SPLIT_VIA_LDS=1) against the ds_bpermute exchange of rounds 1 - 4 (lib_ab/libflucoma_hip_xl0.so,
# built by hand: kernels_stft2.hip with -DFLUHIP_SPLIT_VIA_LDS=0 linked with the A/B build's other objects), alternating
cd "$(dirname "$0")/../../.." || exit 1
for rep in 1 2 3; do
  for lib in ab xl0; do
    r=$(FLUHIP_LIB=flucoma-core_amd/lib_ab/libflucoma_hip_$lib.so python tools/bench_configs.py c5 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print(round(j['ms'],3), round(j['kernel_ms']['features'],3))")
    echo "c5 lib=$lib: ms per call, feature kernel ms: $r"
  done
done
