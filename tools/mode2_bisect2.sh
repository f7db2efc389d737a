#!/bin/bash
# second round of the MODE-2 bisect: which step of the W update is wrong (buffer lengths of 1, 2, 6, 7, 35 four-frame steps),
# every wait conservative at once (FLUHIP_M2_DBG=27), another strip width (fft 1024: 4 groups per strip)
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/mode2
mkdir -p $OUT
AB=flucoma-core_amd/lib_ab
run() { # tag lib mode rank extra...
  local tag=$1 lib=$2 mode=$3 rank=$4; shift 4
  FLUHIP_LIB=$lib FLUHIP_K5_MODE=$mode FLUHIP_K5_MODE_ANY=1 timeout 400 python tools/mode2_bisect.py --tag $tag --rank $rank "$@" > $OUT/$tag.log 2>&1
  echo "rc=$? $(grep SUMMARY $OUT/$tag.log)"
}
for n in 1536 3584 11776 13824 70000; do
  run ref_n$n $AB/libflucoma_hip_ab.so 1 32 --samples $n --cases w_only --repeats 1 --save $OUT/ref_n$n.npz
  run m2_n$n $AB/libflucoma_hip_ab.so 2 32 --samples $n --cases w_only --repeats 2 --ref $OUT/ref_n$n.npz
  run m2dbg27_n$n $AB/libflucoma_hip_m2dbg27.so 2 32 --samples $n --cases w_only --repeats 2 --ref $OUT/ref_n$n.npz
done
run ref_fft1024 $AB/libflucoma_hip_ab.so 1 32 --samples 35000 --fft 1024 --cases w_only,full --repeats 1 --save $OUT/ref_fft1024.npz
run m2_fft1024 $AB/libflucoma_hip_ab.so 2 32 --samples 35000 --fft 1024 --cases w_only,full --repeats 2 --ref $OUT/ref_fft1024.npz
run ref_r16 $AB/libflucoma_hip_ab.so 1 16 --samples 70000 --cases w_only,full --repeats 1 --save $OUT/ref_r16.npz
run m2_r16 $AB/libflucoma_hip_ab.so 2 16 --samples 70000 --cases w_only,full --repeats 2 --ref $OUT/ref_r16.npz
rm -f $OUT/*.npz
