#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/pmcs; rm -rf $out; mkdir -p $out
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $out -o $name -- python bench.py --steps 1 --warmup 0 --iters 2 --no-cpu-baseline > $out/$name.log 2>&1; }
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
run b SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU
run c FETCH_SIZE
run d WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmcs/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        if "stft" in k: print(k, {c: round(sum(v)/len(v)) for c, v in cs.items()})
PY
