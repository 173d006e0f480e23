#!/bin/bash
# A/B of the persistent decode engine's step time (run on the GPU box from the repo root): the in-tree library against another
# build of it (default tools/_bin/libgsv_base.so, copied there before a change), five alternating pairs of tools/mega_prof.py with
# the stamps off.  Run-to-run spread on one box is ~±5 % (two modes ~6 % apart), so only a consistent sign over the pairs counts.
BASE=${1:-$PWD/tools/_bin/libgsv_base.so}
export PROF_OFF=1
for i in 1 2 3 4 5; do
  a=$(timeout -k 10 100 python3 tools/mega_prof.py 2>/dev/null | grep mode | sed 's/.*= \([0-9.]*\) us\/step.*/\1/') || exit 1
  b=$(GSV_LIB_PATH=$BASE timeout -k 10 100 python3 tools/mega_prof.py 2>/dev/null | grep mode | sed 's/.*= \([0-9.]*\) us\/step.*/\1/') || exit 1
  echo "new $a base $b"
done
