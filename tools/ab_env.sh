#!/bin/bash
# usage: tools/ab_env.sh "<bench args>" "ENV1=a ENV2=b" "ENV1=c" ...   (GPU box) the same bench under several environment settings, twice round-robin:
# ms per forces-only step, ms per derivative step, pair-kernel ms (eager stamped steps), reciprocal ms
ARGS=$1; shift
for rep in ${REPS:-1 2}; do for E in "$@"; do
  env $E timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-double $ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-60s' % '$E', d['ms_per_step'], d.get('ms_per_step_with_derivatives'), d.get('ms_per_step_forces_only'), d['roofline']['avg_launch_ms'], d['config'].get('reciprocal_ms'))"
done; done
