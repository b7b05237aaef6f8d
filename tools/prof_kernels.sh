#!/bin/bash
# usage: tools/prof_kernels.sh <tag> [bench args...]   (run on the GPU box; prints per-kernel averages > 1 %)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -- python3 bench.py --no-cpu-baseline --no-double "$@" > gpurun_out/$tag.log 2>&1
python3 - "$tag" <<'PY'
import csv,glob,sys
f=glob.glob("gpurun_out/%s/*/*kernel_stats.csv"%sys.argv[1])[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"])>1.0: print(r["Name"][:64], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
grep "^{" gpurun_out/$tag.log | cut -c1-150
