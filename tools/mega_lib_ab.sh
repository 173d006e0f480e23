#!/bin/bash
# A/B of the persistent decode engine's step time between two builds of the library (alternating fresh processes).
# Usage: tools/mega_lib_ab.sh <libA.so> <libB.so> [pairs] ; PROF_B selects the batch
export PROF_OFF=1
A="$1"; B="$2"; N=${3:-5}
for i in $(seq 1 $N); do
  a=$(GSV_LIB_PATH=$A timeout -k 10 100 python3 tools/mega_prof.py 2>/dev/null | grep mode | sed 's/.*= \([0-9.]*\) us\/step.*/\1/') || exit 1
  b=$(GSV_LIB_PATH=$B timeout -k 10 100 python3 tools/mega_prof.py 2>/dev/null | grep mode | sed 's/.*= \([0-9.]*\) us\/step.*/\1/') || exit 1
  echo "A $a   B $b"
done
