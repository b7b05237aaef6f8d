#!/bin/bash
# GPU box: replay of the compiled pair-step body (tools/ubench_step4.hip), plain and under one --pmc pass
cd "$GRAFT_REPO_ROOT" && hipcc -O3 --offload-arch=gfx950 tools/ubench_step4.hip -o /tmp/ub4 2>/dev/null || exit 1
timeout -k 10 120 /tmp/ub4 > gpurun_out/ub4.txt 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d gpurun_out/ub4pmc -- /tmp/ub4 > gpurun_out/ub4pmc.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/ub4pmc/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:20] + " grid " + r.get("Grid_Size", "?")
    a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
with open("gpurun_out/ub4.txt", "a") as out:
    for k, v in acc.items():
        out.write("%s %s\n" % (k, {c: round(a[0] / a[1]) for c, a in v.items()}))
PY
cat gpurun_out/ub4.txt
