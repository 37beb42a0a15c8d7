#!/bin/bash
# Rank-0 shares (G = 1, 2, 4, 8) of a workload under several knob settings: bash tools/knob_shards.sh c3 "HIPEXEC_LANES=4 HIPEXEC_OVERLAP=0" "..."
WL=$1; shift
for K in "$@"; do
  echo "== $K"
  env $K timeout -k 10 120 python tools/shard_times.py $WL 2>&1 | grep share
done
