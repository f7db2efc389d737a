#!/bin/bash
# config 3 with the finalize launch's rows per workgroup (FLUHIP_FIN_BATCH=4: 512 statistics records per channel, as until
# round 5; 16: 128 records) -- the bench line's us per iteration alternating, then rocprofv3 kernel stats of each
cd "$(dirname "$0")/../../.." || exit 1
export TMPDIR=/tmp FLUHIP_AB=1
out=gpurun_out/c3fin; mkdir -p $out
one() { python tools/bench_configs.py c3 --no-cpu 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print(round(j['us_per_iteration'],1), j['kernel_ms_per_iteration'])"; }
for rep in 1 2; do
  for b in 4 16 0; do echo "c3 batch=$b: $(FLUHIP_FIN_BATCH=$b one)"; done
done
for b in 4 0; do
  d=$out/ks$b; rm -rf $d
  FLUHIP_FIN_BATCH=$b timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o ks -- python tools/bench_configs.py c3 --no-cpu > $out/ks$b.log 2>&1
  find $d -name '*kernel_stats.csv' -exec cp {} $out/c3_batch${b}_kernel_stats.csv \;
  rm -rf $d
  echo "== batch $b"; head -12 $out/c3_batch${b}_kernel_stats.csv | cut -c1-60,150-260 | awk -F, '{print $1, $(NF-6), $(NF-4)}' 
done
python -m pytest tests -x -q -m gpu -k "c3 or wide or split or list or ragged or variants" 2>&1 | tail -3
