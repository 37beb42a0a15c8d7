#!/bin/bash
# Frames-in-flight throughput of a workload's shares under several knob settings:
# bash tools/knob_pipe.sh c2 "A=0" "HIPEXEC_OVERLAP=0" ...
WL=$1; shift
for K in "$@"; do
  echo "== $K"
  env $K timeout -k 10 200 python tools/pipeline_times.py $WL 2 2>&1 | grep "in flight"
done
