#!/bin/bash
# step time of the persistent decode engine for a list of GSV_MEGA_HINT values (alternating fresh processes, N rounds).
# Usage: tools/mega_env_sweep.sh "<v1> <v2> ..." [rounds] ; PROF_B selects the batch.  Run on the GPU box.
export PROF_OFF=1
VALS="$1"; N=${2:-3}
for i in $(seq 1 $N); do
  line=""
  for v in $VALS; do
    t=$(GSV_MEGA_HINT=$v timeout -k 10 100 python3 tools/mega_prof.py 2>/dev/null | grep mode | sed 's/.*= \([0-9.]*\) us\/step.*/\1/') || exit 1
    line="$line  [$v] $t"
  done
  echo "$line"
done
