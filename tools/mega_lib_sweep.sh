#!/bin/bash
# step time of the persistent decode engine for several builds of the library (alternating fresh processes, N rounds).
# Usage: tools/mega_lib_sweep.sh "<libA.so> <libB.so> ..." [rounds] ; PROF_B selects the batch.  Run on the GPU box.
export PROF_OFF=1
LIBS="$1"; N=${2:-3}
for i in $(seq 1 $N); do
  line=""
  for l in $LIBS; do
    t=$(GSV_LIB_PATH=$l timeout -k 10 100 python3 tools/mega_prof.py 2>/dev/null | grep mode | sed 's/.*= \([0-9.]*\) us\/step.*/\1/') || exit 1
    line="$line  [$(basename $l .so | sed 's/libgsv_//')] $t"
  done
  echo "$line"
done
