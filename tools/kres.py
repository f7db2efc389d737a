#!/usr/bin/env python3
"""Register / spill / LDS summary of every kernel in one .hip source (hipcc -Rpass-analysis).
usage: tools/kres.py flucoma-core_amd/csrc/kernels_nmf5.hip [filter-substring]"""
import re, subprocess, sys, os
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950"]
if "nmf" in os.path.basename(src):
    flags.append("-fno-honor-nans")
r = subprocess.run(["/opt/rocm/bin/hipcc", *flags, "-c", src, "-o", "/tmp/_kres.o",
                    "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
cur = None
rows = {}
for line in r.stderr.splitlines():
    m = re.search(r"remark: .*?(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|SGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        m2 = re.search(r"Function Name: (\S+)", line)
        if m2:
            cur = m2.group(1); rows[cur] = {}
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = v; rows[cur] = {}
    elif cur:
        rows[cur][k] = v
for name, d in rows.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*", "", dem).replace("void fluhip::", "")
    if flt in dem:
        print(f"{dem:55s} vgpr {d.get('VGPRs','?'):>4} agpr {d.get('AGPRs','?'):>4} spill {d.get('VGPRs Spill','?'):>3} scratch {d.get('ScratchSize [bytes/lane]','?'):>4}")
