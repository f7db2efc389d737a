"""Per-instantiation register / spill table of a -Rpass-analysis=kernel-resource-usage compile log of kernels_nmf5*.hip.
    hipcc ... -Rpass-analysis=kernel-resource-usage -c csrc/kernels_nmf5_off.hip -o /tmp/x.o 2> log;  python tools/k5_resources.py log
Template arguments print as <M,NG,NS,WPS,INSTR,MODE,DS,LIST,SIDEQ,KPM>."""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
rows = []
for b in txt.split("Function Name: ")[1:]:
    name = b.split()[0]
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    m = re.search(r"<(.*)>", dem)

    def g(k):
        r = re.search(k + r": (\d+)", b)
        return int(r.group(1)) if r else -1
    rows.append((m.group(1) if m else dem, g("VGPRs"), g("AGPRs"), g(r"VGPRs Spill"), g(r"SGPRs Spill"), g(r"ScratchSize \[bytes/lane\]")))
print("%-40s %5s %5s %6s %6s %8s" % ("<M,NG,NS,WPS,INSTR,MODE,DS,LIST,SIDEQ,KPM>", "VGPR", "AGPR", "vspill", "sspill", "scratch"))
for r in sorted(rows, key=lambda r: [int(v) for v in re.findall(r"-?\d+", r[0])]):
    print("%-40s %5d %5d %6d %6d %8d%s" % (*r, "   <-- SPILLS" if r[3] > 0 or r[5] > 0 else ""))
