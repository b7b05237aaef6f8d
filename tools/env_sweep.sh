#!/bin/bash
# usage: tools/env_sweep.sh <tag> "<ENV=.. ENV=..>" ["<ENV..>" ...] -- [bench args]   (GPU box) one bench line per environment setting: step times and per-kernel PME stamps
TAG=$1; shift
SETS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do SETS+=("$1"); shift; done
[ "$1" == "--" ] && shift
mkdir -p gpurun_out
for S in "${SETS[@]}"; do
  env $S python3 bench.py --no-cpu-baseline --no-double "$@" 2>gpurun_out/sweep_$TAG.err | tail -1 > gpurun_out/sweep_$TAG.json && python3 tools/pmeprint.py gpurun_out/sweep_$TAG.json "[$S]" | tee -a gpurun_out/sweep_$TAG.txt || { echo "[$S] failed"; tail -3 gpurun_out/sweep_$TAG.err; }
done
