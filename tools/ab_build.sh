#!/bin/bash
# A/B of the two hierarchy builds on whole frames as bench.py renders them: hipexec_create's Morton-order tree
# (--fast-build) against the surface-area topology of hipexec_optimize (the default of bench.py).
#   bash tools/ab_build.sh "<workloads>"     (GPU box, repository root)
WLS=${1:-c2}
for round in 1 2; do
  for how in --fast-build ""; do
    for wl in $WLS; do
      echo -n "build=${how:-optimized} "
      bash tools/bench_reps.sh "$wl" 1 $how || exit 1
    done
  done
done
