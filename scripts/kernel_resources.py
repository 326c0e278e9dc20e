"""Summarise hipcc -Rpass-analysis=kernel-resource-usage remarks: python scripts/kernel_resources.py remarks.txt [filter]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].strip()
    try:
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    except Exception:
        pass
    if flt not in name:
        continue
    g = lambda k: (re.search(k + r": (\d+)", b) or [0, "?"])[1]
    print(name.replace("igx::", "")[:110], "| VGPR", g("VGPRs"), "AGPR", g("AGPRs"), "scratch", g(r"ScratchSize \[bytes/lane\]"),
          "occ", g(r"Occupancy \[waves/SIMD\]"), "sgprspill", g("SGPRs Spill"), "vgprspill", g("VGPRs Spill"))
