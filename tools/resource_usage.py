"""Per-kernel register / LDS / scratch table from `hipcc -Rpass-analysis=kernel-resource-usage` output.
Usage: python tools/resource_usage.py remarks.txt [name filter ...]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
filters = sys.argv[2:]
rows = []
for m in re.finditer(r"Function Name: (\S+)", txt):
    seg = txt[m.end():m.end() + 2500]
    def g(k):
        mm = re.search(re.escape(k) + r": (\w+)", seg)
        return mm.group(1) if mm else "?"
    rows.append((m.group(1), g("VGPRs"), g("AGPRs"), g("ScratchSize [bytes/lane]"), g("Occupancy [waves/SIMD]"), g("VGPRs Spill"), g("LDS Size [bytes/block]")))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
print("%-120s %5s %5s %7s %4s %6s %7s" % ("kernel", "VGPR", "AGPR", "scratch", "occ", "vspill", "LDS"))
for r, n in zip(rows, names):
    n = n.replace("snb::", "")
    if filters and not any(f in n for f in filters):
        continue
    print("%-120s %5s %5s %7s %4s %6s %7s" % (n[:120], *r[1:]))
