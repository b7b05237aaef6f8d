#!/bin/bash
# usage: tools/ab_pme.sh <tag> <libA.so> <libB.so> [bench args]   (GPU box) A/B/A/B of two builds through SNB_LIB_PATH: step times and per-kernel PME stamps
TAG=$1; A=$2; B=$3; shift; shift; shift
mkdir -p gpurun_out
for rep in 1 2; do for L in "$A" "$B"; do
  SNB_LIB_PATH=$L python3 bench.py --no-cpu-baseline --no-double "$@" 2>/dev/null | tail -1 > gpurun_out/ab_$TAG.json && python3 tools/pmeprint.py gpurun_out/ab_$TAG.json $(basename $L) | tee -a gpurun_out/ab_$TAG.txt
done; done
