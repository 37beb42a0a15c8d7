#!/bin/bash
# A/B of builds of the library on whole frames as bench.py renders them (frames in flight):
#   bash tools/ab_bench_libs.sh "<workloads>" build_ab/a.so build_ab/b.so [...]
# Alternates the builds (a b a b) and prints ms per frame and the per-class kernel times of each run.
set -e
WLS=${1:-c2}
shift
LIB=raytracer_amd/lib/libhip_executor.so
cp "$LIB" /tmp/ab_keep.so
trap 'cp /tmp/ab_keep.so "$LIB"' EXIT
for round in 1 2; do
    for so in "$@"; do
        cp "$so" "$LIB"
        for wl in $WLS; do
            echo -n "$(basename "$so") "
            bash tools/bench_reps.sh "$wl" 1 || exit 1
        done
    done
done
