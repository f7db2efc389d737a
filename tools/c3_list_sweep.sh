export FLUHIP_AB=1   # the build whose experiment switches are live (flucoma-core_amd/build.py --ab)
# c3 (2 x 10 min, rank 128) under schedule switches: bash tools/c3_list_sweep.sh
run() { echo "== $*"; env "$@" python tools/bench_configs.py c3 --no-cpu 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['us_per_iteration'],1),'us/it frac',round(d['roofline']['frac'],4), d['kernel_ms_per_iteration'], {k:(round(v['cycles_per_launch']),round(v['sustained_mhz'])) for k,v in d['update_clocks'].items()}, d['schedule'])"; }
run A=1
run FLUHIP_NO_SIDE=1
run A=1
run FLUHIP_NO_SIDE=1
