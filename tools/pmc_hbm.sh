#!/bin/bash
# usage: tools/pmc_hbm.sh <tag> [bench args...]   (GPU box)
# HBM-side traffic of every kernel of the bench step, collected as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in
# SEPARATE --pmc passes (they do not fit one pass), only --kernel-trace beside them.  Summary -> gpurun_out/<tag>_hbm.json
tag=$1; shift
export SNB_OVERLAP=0 SNB_SIDE_REBUILD=0      # kernels alone: one pair-kernel launch per step, counters per launch = per step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/${tag}_$c -- python3 bench.py --no-cpu-baseline --no-double "$@" > gpurun_out/${tag}_$c.log 2>&1 || exit 1
done
python3 - "$tag" <<'PY'
import csv, glob, json, sys, collections
tag = sys.argv[1]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/%s_%s/*/*counter_collection.csv" % (tag, c))[0]
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        k = r["Kernel_Name"].split("(")[0]
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    for k, (v, n) in acc.items():
        out.setdefault(k, {})[c + "_per_launch_raw"] = v / n
        out[k]["launches_" + c] = n
json.dump(out, open("gpurun_out/%s_hbm.json" % tag, "w"), indent=1, sort_keys=True)
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("FETCH_SIZE_per_launch_raw", 0))[:14]:
    print(k[:60], {a: round(b, 1) for a, b in v.items()})
PY
