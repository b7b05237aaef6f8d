#!/bin/bash
# GPU box: packed-fp32 issue microbenchmark, plain and under one --pmc pass (how SQ_INSTS_VALU counts packed instructions)
cd "$GRAFT_REPO_ROOT" && hipcc -O3 --offload-arch=gfx950 -Wno-unused-result tools/ubench_valu3.hip -o /tmp/ub3 2>/dev/null || exit 1
timeout -k 10 120 /tmp/ub3 > gpurun_out/ub3.txt 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d gpurun_out/ub3pmc -- /tmp/ub3 > gpurun_out/ub3pmc.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/ub3pmc/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:40] + " grid " + r.get("Grid_Size", "?")
    a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
with open("gpurun_out/ub3.txt", "a") as out:
    for k, v in acc.items():
        out.write("%s %s\n" % (k, {c: round(a[0] / a[1]) for c, a in v.items()}))
PY
cat gpurun_out/ub3.txt
