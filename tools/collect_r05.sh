#!/bin/bash
# round 5 records (run on the GPU box, copies go to profiles/r05/): the bench line, the rocprofv3 kernel stats of the same
# command, the PMC passes of the update kernel (tied to the kernel source by its hash), and the per-config sets for c2, c3, c5
export TMPDIR=/tmp; out=gpurun_out/r05final; tag=${1:-v1}; mkdir -p $out
python bench.py > $out/bench_$tag.json 2> $out/bench_$tag.err
d=$out/ks; rm -rf $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o b -- python bench.py --no-cpu-baseline --configs none > $out/ks.log 2>&1
find $d -name '*kernel_stats.csv' -exec cp {} $out/bench_${tag}_kernel_stats.csv \;
rm -rf $out/pmc; bash tools/pmc_bench.sh $out/pmc --configs none > $out/pmc_passes.txt 2>&1
python tools/pmc_summary.py $out/pmc nmf_update5_kernel $out/pmc_update_kernel.json > /dev/null
bash tools/profile_configs.sh $out/cfg c2 c3 c5 > $out/cfg.log 2>&1
rm -rf $d
ls $out $out/cfg
