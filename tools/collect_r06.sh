#!/bin/bash
# round 6 records (run on the GPU box; copies go to profiles/r06/): the bench line, the rocprofv3 kernel stats of the same command,
# the PMC passes of the update kernel (tied to the kernel source by its hash), the perf matrix, the rocprofv3 kernel stats of the
# ragged 64 x 40 corpus, and the per-config sets for c2, c3, c5
export TMPDIR=/tmp; tag=${1:-final}; out=${2:-gpurun_out/r06final}; mkdir -p $out
python bench.py > $out/bench_$tag.json 2> $out/bench_$tag.err
d=$out/ks; rm -rf $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o b -- python bench.py --no-cpu-baseline --configs none > $out/ks.log 2>&1
find $d -name '*kernel_stats.csv' -exec cp {} $out/bench_${tag}_kernel_stats.csv \;
rm -rf $d
rm -rf $out/pmc; bash tools/pmc_bench.sh $out/pmc --configs none > $out/pmc_passes.txt 2>&1
python tools/pmc_summary.py $out/pmc nmf_update5_kernel $out/pmc_update_kernel.json > /dev/null
rm -rf $out/pmc
python tools/perf_matrix.py --out $out/perf_matrix_$tag.json > /dev/null 2> $out/perf_matrix_$tag.err
d=$out/rag; rm -rf $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o rag -- python tools/ragged_timing.py 64 40 4 16 32 ragged > $out/ragged_timing.json 2> $out/rag.err
find $d -name '*kernel_stats.csv' -exec cp {} $out/ragged_64x40_kernel_stats.csv \;
rm -rf $d
bash tools/profile_configs.sh $out/cfg c2 c3 c5 > $out/cfg.log 2>&1
ls $out $out/cfg
