#!/bin/bash
# counter passes over the STFT phase of the bench shard (tools/stft_timing.py, first shape); separate --pmc passes
# usage: tools/pmc_stft.sh <outdir>
export TMPDIR=/tmp
out=${1:-gpurun_out/pmcs}; rm -rf $out; mkdir -p $out
export STFT_SHAPES=${STFT_SHAPES:-0}
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $out -o $name -- python tools/stft_timing.py > $out/$name.log 2>&1; }
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
run b SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU
run c FETCH_SIZE
run d WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run e SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_LDS SQ_INSTS_FLAT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
run f TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
python - $out <<'PY'
import csv, glob, collections, sys
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        if "stft" in k or "transpose" in k: print(k, {c: round(sum(v)/len(v)) for c, v in cs.items()})
PY
