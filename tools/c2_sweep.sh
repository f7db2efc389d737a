export FLUHIP_AB=1   # the build whose experiment switches are live (flucoma-core_amd/build.py --ab)
for cfg in "X=1" "FLUHIP_K5_WPS=2" "FLUHIP_PLAN_W=33" "FLUHIP_PLAN_W=17" "FLUHIP_PLAN_W=13" "FLUHIP_PLAN_SPLIT=23" "FLUHIP_PLAN_SPLIT=64" "FLUHIP_K5_MODE=2" "FLUHIP_K5_MODE=0"; do
  echo "== $cfg: $(env $cfg python tools/bench_configs.py c2 --no-cpu 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(round(r['us_per_iteration'],1), r['kernel_ms_per_iteration'], r['schedule'])")"
done
