#!/bin/bash
# usage: tools/timeline_run.sh <tag> "<ENV=... settings>" [bench args]   (GPU box) kernel-trace of a short bench run under the given environment,
# then the timeline of one replayed step and a one-line-per-step table (tools/step_timeline.py) into gpurun_out/<tag>.txt
tag=$1; envs=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export $envs
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$tag -- python3 bench.py --no-cpu-baseline --no-double --steps 40 --warmup 5 "$@" > gpurun_out/$tag.log 2>&1
python3 tools/step_timeline.py gpurun_out/$tag 12 1 > gpurun_out/$tag.txt
python3 tools/step_timeline.py gpurun_out/$tag summary 100 100 >> gpurun_out/$tag.txt
rm -rf gpurun_out/$tag
echo "== $tag: $envs"; head -14 gpurun_out/$tag.txt
