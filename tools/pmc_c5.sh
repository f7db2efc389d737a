#!/bin/bash
# config 5's fused feature kernel under the SQ counters that say where a wavefront's cycles go (separate --pmc passes, no tracing)
export TMPDIR=/tmp; out=${1:-gpurun_out/pmc_c5}; rm -rf $out; mkdir -p $out
pass() { name=$1; shift; timeout 600 rocprofv3 --pmc "$@" --output-format csv -d $out -o $name -- python tools/bench_configs.py c5 --no-cpu > $out/$name.log 2>&1; }
pass p1 SQ_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
pass p2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU
pass p3 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32
pass p4 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL
pass p5 SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM
python - "$out" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in sorted(glob.glob(out + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "stft_feat_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
# the big launches only (the tool also runs a 64-slice warm-up call)
res = {}
for k, v in agg.items():
    big = [x for x in v if x >= 0.5 * max(v)] if max(v) > 0 else v
    res[k] = sum(big) / len(big)
frames = 8192 * 173
print(json.dumps({k: round(v, 1) for k, v in sorted(res.items())}, indent=0))
print("per frame:", {k: round(v / frames, 2) for k, v in sorted(res.items()) if k.startswith("SQ_INSTS")})
json.dump(res, open(out + "/summary.json", "w"), indent=1)
PY
