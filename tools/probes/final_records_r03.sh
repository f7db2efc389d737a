out=gpurun_out/final; tag=v5; export TMPDIR=/tmp; mkdir -p $out
timeout 400 python bench.py > $out/bench_$tag.json 2> $out/bench_$tag.err
rm -rf $out/_ks; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/_ks -o ks -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/_ks.log 2>&1
find $out/_ks -name '*kernel_stats.csv' -exec cp {} $out/bench_${tag}_kernel_stats.csv \; ; rm -rf $out/_ks $out/_ks.log
FLUHIP_BENCH_BACKEND=nccl timeout 400 python bench.py --no-cpu-baseline > $out/bench_${tag}_one_rank_rccl.json 2> $out/rccl.err
timeout 600 python bench.py --buffers 1024 --steps 1 --warmup 1 --no-cpu-baseline > $out/bench_${tag}_1024_buffers_one_gpu.json 2> $out/1024.err
for r in 8 16 64 128; do timeout 300 python bench.py --rank $r --iters 50 --steps 2 --warmup 1 --no-cpu-baseline; done > $out/bench_${tag}_other_ranks.jsonl 2> $out/ranks.err
find $out -name '*.err' -size 0 -delete
for f in $out/bench_$tag.json $out/bench_${tag}_one_rank_rccl.json $out/bench_${tag}_1024_buffers_one_gpu.json; do python -c "
import json,sys; d=json.loads(open('$f').readline()); print('$f', round(d['value']), round(d['ms_per_step'],1), d['roofline']['frac'], d['roofline'].get('cycles_per_launch'), d.get('roofline_stft',{}).get('frac'))"; done
python -c "
import json
for l in open('$out/bench_${tag}_other_ranks.jsonl'): d=json.loads(l); print(d['config'].get('rank'), round(d['value']), round(d['ms_per_step'],2), round(d['roofline']['frac'],4), d['roofline']['bound'])"
head -4 $out/bench_${tag}_kernel_stats.csv | cut -c1-150
