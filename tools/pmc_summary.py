#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of tools/pmc_bench.sh for one kernel into the JSON bench.py quotes.
usage: tools/pmc_summary.py <pmc dir> <kernel-name substring> <out.json>"""
import collections, csv, glob, hashlib, json, os, sys
d, sub, outp = sys.argv[1:4]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KSRC = os.path.join(ROOT, "flucoma-core_amd", "csrc", "kernels_nmf5.hip")
agg = collections.defaultdict(list)
for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
B, T, F, K = 128, 862, 1025, 32
alg = (F * T + 2 * (F * K + K * T)) * 8 * B
out = {
    "source": "rocprofv3 --pmc, separate passes (tools/pmc_bench.sh), MI355X, bench.py workload (128 x (862 x 1025), K = 32)",
    "kernel": sub + " (one factor update of 128 buffers)",
    # the counters describe THIS source state of the kernel: bench.py quotes `traffic` only while the file still hashes to it
    "kernel_source": "flucoma-core_amd/csrc/kernels_nmf5.hip",
    "kernel_source_sha256": hashlib.sha256(open(KSRC, "rb").read()).hexdigest(),
    "launches_averaged": {k: len(v) for k, v in agg.items()},
    "FETCH_SIZE_KB_raw": m.get("FETCH_SIZE"),
    "fetch_correction": "x2: gfx950 rocprofv3 reports half the bytes of 16-byte-per-lane streaming loads "
                        "(MI355X_MICROARCH.md, HBM section); cross-check: TCC_MISS_sum x 128 B",
    "WRITE_SIZE_KB": m.get("WRITE_SIZE"),
    "hbm_bytes_per_launch": (2 * m.get("FETCH_SIZE", 0) + m.get("WRITE_SIZE", 0)) * 1024,
    "tcc_miss_bytes_per_launch": m.get("TCC_MISS_sum", 0) * 128,
    "algorithmic_bytes_per_launch": alg,
    "TCC_HIT_sum": m.get("TCC_HIT_sum"), "TCC_MISS_sum": m.get("TCC_MISS_sum"),
    "SQ_WAVES": m.get("SQ_WAVES"),
    "SQ_VALU_MFMA_BUSY_CYCLES": m.get("SQ_VALU_MFMA_BUSY_CYCLES"),
    "SQ_BUSY_CYCLES": m.get("SQ_BUSY_CYCLES"), "GRBM_GUI_ACTIVE": m.get("GRBM_GUI_ACTIVE"),
    "SQ_WAVE_CYCLES_quad": m.get("SQ_WAVE_CYCLES"), "SQ_WAIT_ANY_quad": m.get("SQ_WAIT_ANY"),
    "SQ_WAIT_INST_ANY_quad": m.get("SQ_WAIT_INST_ANY"), "SQ_ACTIVE_INST_ANY_quad": m.get("SQ_ACTIVE_INST_ANY"),
    "SQ_INSTS_VALU": m.get("SQ_INSTS_VALU"), "SQ_INSTS_LDS": m.get("SQ_INSTS_LDS"),
    "SQ_INSTS_VMEM_RD": m.get("SQ_INSTS_VMEM_RD"), "SQ_INSTS_VALU_MFMA_MOPS_F64": m.get("SQ_INSTS_VALU_MFMA_MOPS_F64"),
}
if m.get("SQ_VALU_MFMA_BUSY_CYCLES") and m.get("SQ_WAVE_CYCLES"):
    # SQ_WAVE_CYCLES counts in quads (4 cycles) summed over wavefronts; MFMA busy in cycles summed over SIMDs
    out["mfma_busy_fraction"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * m["SQ_WAVE_CYCLES"])
json.dump(out, open(outp, "w"), indent=1)
print(json.dumps(out, indent=1))
