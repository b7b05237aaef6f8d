#!/bin/bash
# usage: tools/pmc_sq.sh <tag> "<counters>" [bench args...]   (GPU box) one --pmc pass, per-kernel averages
tag=$1; ctrs=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d gpurun_out/$tag -- python3 bench.py --no-cpu-baseline --no-double "$@" > gpurun_out/$tag.log 2>&1 || exit 1
python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
f = glob.glob("gpurun_out/%s/*/*counter_collection.csv" % tag)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:60]
    a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, v in acc.items():
    if any(x in k for x in ("k_direct", "k_spread", "k_interp", "k_conv", "k_fft", "k_plane", "k_nbBuild")):
        print(k, {c: round(a[0] / a[1]) for c, a in v.items()})
PY
