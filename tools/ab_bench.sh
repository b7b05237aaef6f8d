#!/bin/bash
# usage: tools/ab_bench.sh <libA.so> <libB.so> [bench args]   (GPU box) A/B/A/B of two builds of the engine through SNB_LIB_PATH: ms per step, pair-kernel ms
A=$1; B=$2; shift; shift
for rep in 1 2; do for L in "$A" "$B"; do
  SNB_LIB_PATH=$L python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L'.split('/')[-1], d['ms_per_step'], d['ms_per_step_with_derivatives'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
done; done
