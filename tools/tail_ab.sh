export FLUHIP_AB=1   # the build whose experiment switches are live (flucoma-core_amd/build.py --ab)
# config 3 with the H update as one launch (FLUHIP_TAIL_SPLIT=0) against two (api.hip plan_tail; =3 / =5 force the pieces), alternating on one box
set -x
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_random_shapes.py -m gpu -q -k two_launch 2>&1 | tail -5
for v in "" "FLUHIP_TAIL_SPLIT=0" "FLUHIP_TAIL_SPLIT=3" "FLUHIP_TAIL_SPLIT=5" "FLUHIP_TAIL_SPLIT=0" ""; do
  echo "== tail: $v"
  env $v timeout 300 python tools/bench_configs.py c3 --no-cpu 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print(round(d['us_per_iteration'],1), 'us/it frac', round(d['roofline']['frac'],4), d['kernel_ms_per_iteration'], {k:(int(v['cycles_per_launch']), int(v['sustained_mhz'])) for k,v in d['update_clocks'].items()})
"
done
