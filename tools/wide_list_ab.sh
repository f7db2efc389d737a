export FLUHIP_AB=1   # the build whose experiment switches are live (flucoma-core_amd/build.py --ab)
for K in 64 128; do for v in 0 1; do
echo "K=$K FLUHIP_LIST_PLAN=$v: $(FLUHIP_LIST_PLAN=$v timeout 200 python tools/batch_timing.py 128 1.6 $K 8 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print({k: (round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k in ('stft_ms','nmf0_ms','nmf_ms','nmf_progress_ms','us_per_iteration','kernel_ms','plan')})")"
done; done
