#!/bin/bash
# One gpurun call of the MODE-2 bisect (tools/mode2_bisect.py): the production form as reference, the in-place form forced onto
# rank 32 (FLUHIP_K5_MODE=2 FLUHIP_K5_MODE_ANY=1) in the plain A/B build and in the -DFLUHIP_M2_DBG=<bits> builds given as
# arguments (default: 1 2 4 24; build.py build_m2dbg makes them), then ranks 64 / 128 -- where the form is production -- in the
# production build with every replica compared.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/mode2_bisect.sh [bits ...]'
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/mode2
mkdir -p $OUT
AB=flucoma-core_amd/lib_ab
BITS=${*:-1 2 4 24}
SAMPLES=${SAMPLES:-70000}
run() { # tag lib mode rank extra...
  local tag=$1 lib=$2 mode=$3 rank=$4; shift 4
  if [ -n "$mode" ]; then
    FLUHIP_LIB=$lib FLUHIP_K5_MODE=$mode FLUHIP_K5_MODE_ANY=1 timeout 400 python tools/mode2_bisect.py --tag $tag --rank $rank --samples $SAMPLES "$@" > $OUT/$tag.log 2>&1
  else
    FLUHIP_LIB=$lib timeout 400 python tools/mode2_bisect.py --tag $tag --rank $rank --samples $SAMPLES "$@" > $OUT/$tag.log 2>&1
  fi
  echo "rc=$? $(grep SUMMARY $OUT/$tag.log)"
}
run ref32 $AB/libflucoma_hip_ab.so 1 32 --save $OUT/ref32.npz --repeats 1
run m2_plain $AB/libflucoma_hip_ab.so 2 32 --ref $OUT/ref32.npz
for b in $BITS; do
  run m2_dbg$b $AB/libflucoma_hip_m2dbg$b.so 2 32 --ref $OUT/ref32.npz
done
# ranks 64 / 128: MODE 1 (rank 64) and MODE 0 (rank 128) as references for the production in-place form
run ref64 $AB/libflucoma_hip_ab.so 1 64 --save $OUT/ref64.npz --repeats 1
run prod64 flucoma-core_amd/lib/libflucoma_hip.so "" 64 --ref $OUT/ref64.npz
run ref128 $AB/libflucoma_hip_ab.so 0 128 --save $OUT/ref128.npz --repeats 1
run prod128 flucoma-core_amd/lib/libflucoma_hip.so "" 128 --ref $OUT/ref128.npz
rm -f $OUT/*.npz
