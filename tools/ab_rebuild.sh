#!/bin/bash
# usage: tools/ab_rebuild.sh [config] "ENV1=a" "ENV2=b" ...   (GPU box) un-profiled cost of the rebuilds: tools/host_launch_cost.py in its
# `rebuild` mode (300 forces-only steps, a rebuild every REBUILD_EVERY = 20) under each environment setting, three rounds, round-robin
CFG=$1; shift
for rep in 1 2 3; do for E in "$@"; do
  echo "== $E"; env $E SNB_OVERLAP=${SNB_OVERLAP:-1} timeout -k 10 300 python3 tools/host_launch_cost.py $CFG rebuild 2>/dev/null | tail -3
done; done
