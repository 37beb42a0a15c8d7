#!/bin/bash
# A/B of two builds of the library on the GPU box: bash tools/ab_libs.sh "<workloads>" build_ab/a.so build_ab/b.so
# Alternates the builds (a b a b) and prints the G = 1 line of tools/shard_times.py for each workload.
set -e
WLS=${1:-c2}
shift
LIB=raytracer_amd/lib/libhip_executor.so
cp "$LIB" /tmp/ab_keep.so
for round in 1 2; do
    for so in "$@"; do
        cp "$so" "$LIB"
        for wl in $WLS; do
            echo -n "$(basename "$so") "
            timeout -k 10 100 python tools/shard_times.py "$wl" 2>/dev/null | grep -E "${PAT:-G=1:}" || { cp /tmp/ab_keep.so "$LIB"; exit 1; }
        done
    done
done
cp /tmp/ab_keep.so "$LIB"
