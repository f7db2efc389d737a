#!/bin/bash
# counter passes over a short bench run (separate --pmc passes; never combined with tracing)
# usage: tools/pmc_bench.sh <outdir> [extra bench args]
out=$1; shift
export TMPDIR=/tmp
mkdir -p $out
run() { # name, counters...
  name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $out -o $name -- python bench.py --steps 1 --warmup 0 --iters 6 --no-cpu-baseline "${EXTRA[@]}" > $out/$name.log 2>&1
}
EXTRA=("$@")
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run tcc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run sq2 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
python - "$out" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", f)
    for k, cs in agg.items():
        if "nmf_update" in k or "stft" in k or "colnorm" in k:
            print(" ", k, {c: (round(sum(v) / len(v), 1), len(v)) for c, v in cs.items()})
PY
