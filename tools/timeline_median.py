"""Median duration of every kernel class and of the whole step over the replayed derivative steps of a timeline file written by
tools/timeline_run.sh (the one-line-per-step table).  usage: python tools/timeline_median.py gpurun_out/<tag>.txt"""
import re, sys, statistics as st
rows = []
for line in open(sys.argv[1]):
    m = re.match(r"\s*(\d+)\s+([\d.]+) \| (.*)", line)
    if not m:
        continue
    total = float(m.group(2)); cells = [c.strip() for c in m.group(3).split("|")]
    if not any(c.startswith("pairE") for c in cells) or any(c.startswith("nbKeys") for c in cells) or total > 700:
        continue
    d = {"step": total}
    for c in cells:
        name, se = c.rsplit(" ", 1); s, e = se.split("+")
        key = name if name not in d else name + "#2"
        d[key] = float(e); d[key + "@"] = float(s)
    rows.append(d)
keys = ["step", "pairE", "pairE#2", "spreadOwn", "spreadMerg", "planeXY", "fftZInvMix", "interpolat"]
print(len(rows), "steps;", "  ".join("%s %.0f" % (k, st.median([r[k] for r in rows if k in r])) for k in keys if any(k in r for r in rows)))
