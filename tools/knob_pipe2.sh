#!/bin/bash
# Shares with frames in flight under knob settings: bash tools/knob_pipe2.sh c2 "2 4" "HIPEXEC_OVERLAP=0" "HIPEXEC_OVERLAP=0 HIPEXEC_TAIL_MAX=0" ...
WL=$1; KS=$2; shift; shift
for K in "$@"; do
  echo "== $K"
  env $K timeout -k 10 200 python tools/pipeline_times.py $WL $KS 2>&1 | grep "in flight" || exit 1
done
