"""Timeline of replayed steps from a `rocprofv3 --kernel-trace --output-format csv` run of bench.py: for a few steps (one step = from
one k_gatherPositions launch to the next) prints every kernel's start and end relative to the step's start, so that overlap between the
pair kernel's launches and the PME chain can be read directly.
usage: python tools/step_timeline.py <dir with *_kernel_trace.csv> [first step index from the end, default 12] [steps, default 2]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
summary = len(sys.argv) > 2 and sys.argv[2] == "summary"
if summary:
    del sys.argv[2]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 12
count = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?")), r.get("Grid_Size", "?"), r.get("LDS_Block_Size", "?"), r.get("VGPR_Count", "?")) for r in csv.DictReader(open(f))]
rows.sort()
starts = [i for i, r in enumerate(rows) if "k_gatherPositions" in r[2]]
if summary:      # one line per step: length, then (start, duration) of every kernel by short name
    for s in range(max(0, len(starts) - back), min(len(starts) - 1, len(starts) - back + count)):
        a, b = starts[s], starts[s + 1]
        t0 = rows[a][0]
        cells = []
        for r in rows[a:b]:
            name = r[2].split("(")[0].replace("void snb::", "").split("<")[0].replace("k_", "")[:10]
            if "directPacked" in r[2] or "k_direct<" in r[2]:
                name = "pairE" if ("true, true, false" in r[2] or ", true>" in r[2].split("(")[0]) else "pair"
            cells.append("%s %.0f+%.0f" % (name, (r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3))
        print("%4d %6.1f | %s" % (s, (rows[b][0] - t0) / 1e3, " | ".join(cells)))
    sys.exit(0)
for s in range(len(starts) - back, len(starts) - back + count):
    a, b = starts[s], starts[s + 1]
    t0 = rows[a][0]
    print("step %d: %.1f us to the next step's gather" % (s, (rows[b][0] - t0) / 1e3))
    for r in rows[a:b]:
        name = r[2].split("(")[0].replace("void snb::", "")[:60]
        print("   %8.1f %8.1f  %7.1f us  q%-3s grid %-8s lds %-6s vgpr %-4s %s" % ((r[0] - t0) / 1e3, (r[1] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[3], r[4], r[5], r[6], name))

# compact table of every step in a range: python tools/step_timeline.py <dir> summary [first step from the end] [steps]
