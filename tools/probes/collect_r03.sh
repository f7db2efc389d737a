#!/bin/bash
# the round-3 record set: bench line, its rocprofv3 kernel stats and PMC passes, the one-rank RCCL line, the whole
# 1024-buffer corpus of BASELINE config 4 on ONE GPU (strong-scaling reference), per-config sets, ragged / small-batch tools
#   tools/collect_r03.sh <outdir> [tag]
out=$1; tag=${2:-v1}
export TMPDIR=/tmp
mkdir -p "$out"
python bench.py > "$out/bench_$tag.json" 2> "$out/bench_$tag.err"
rm -rf "$out/_ks"; rocprofv3 --kernel-trace --stats --output-format csv -d "$out/_ks" -o ks -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$out/_ks.log" 2>&1
find "$out/_ks" -name '*kernel_stats.csv' -exec cp {} "$out/bench_${tag}_kernel_stats.csv" \; ; rm -rf "$out/_ks" "$out/_ks.log"
bash tools/pmc_bench.sh "$out/pmc_$tag" > "$out/pmc_$tag.log" 2>&1
python tools/pmc_summary.py "$out/pmc_$tag" nmf_update5_kernel "$out/pmc_update_kernel.json" > /dev/null 2>&1
rm -f "$out"/pmc_$tag/*.log "$out"/pmc_$tag/*agent_info.csv
FLUHIP_BENCH_BACKEND=nccl python bench.py --no-cpu-baseline > "$out/bench_${tag}_one_rank_rccl.json" 2> "$out/bench_${tag}_one_rank_rccl.err"
python bench.py --buffers 1024 --steps 1 --warmup 1 --no-cpu-baseline > "$out/bench_${tag}_1024_buffers_one_gpu.json" 2> "$out/bench_${tag}_1024.err"
for r in 8 16 64 128; do python bench.py --rank $r --iters 50 --steps 2 --warmup 1 --no-cpu-baseline; done > "$out/bench_${tag}_other_ranks.jsonl" 2> "$out/bench_${tag}_other_ranks.err"
for B in 1 2 4 8 16 32 64; do python tools/batch_timing.py $B; done > "$out/small_batches_$tag.jsonl" 2>/dev/null
python tools/ragged_timing.py 64 40 4 16 32 both > "$out/ragged_64x40_$tag.json" 2>/dev/null
python tools/ragged_timing.py 256 100 2 20 32 both > "$out/ragged_256x100_$tag.json" 2>/dev/null
find "$out" -name '*.err' -size 0 -delete
