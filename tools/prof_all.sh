#!/bin/bash
# usage: tools/prof_all.sh <tag> [bench args...]   (GPU box) per-kernel averages of EVERY kernel of the bench command, sorted by total time
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -- python3 bench.py --no-cpu-baseline "$@" > gpurun_out/$tag.log 2>&1
python3 - "$tag" <<'PY'
import csv,glob,sys
f=glob.glob("gpurun_out/%s/*/*kernel_stats.csv"%sys.argv[1])[0]
rows=list(csv.DictReader(open(f)))
for r in rows:
    print("%-70s calls %6s avg %8.1f us total %8.2f ms  %5.1f%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, float(r["Percentage"])))
PY
grep "^{" gpurun_out/$tag.log | cut -c1-150
