#!/bin/bash
# One environment knob over several values on whole frames: bash tools/sweep_knob.sh HIPEXEC_SHADOW_ROUND_WAVES "0 4096" "c2 c3" [reps]
export HIPEXEC_KNOBS=1
VAR=$1; VALS=$2; WLS=${3:-c2}; REPS=${4:-1}
for rep in $(seq $REPS); do
  for wl in $WLS; do
    for v in $VALS; do
      export $VAR=$v
      echo -n "$VAR=$v "
      bash tools/bench_reps.sh $wl 1 || exit 1
    done
  done
done
