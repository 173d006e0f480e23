#!/bin/bash
# step time of the persistent decode engine over repeated process starts, for each role mapping (GSV_MEGA_MAP = 2: by the XCD a
# workgroup runs on, 1: by blockIdx % 8): the run-to-run modes of tools/mega_ab.sh, per mapping.  Run on the GPU box.
export PROF_OFF=1
for i in 1 2 3 4 5 6; do
  a=$(GSV_MEGA_MAP=2 timeout -k 10 100 python3 tools/mega_prof.py 2>/dev/null | grep mode | sed 's/.*= \([0-9.]*\) us\/step.*/\1/') || exit 1
  b=$(GSV_MEGA_MAP=1 timeout -k 10 100 python3 tools/mega_prof.py 2>/dev/null | grep mode | sed 's/.*= \([0-9.]*\) us\/step.*/\1/') || exit 1
  echo "xcc-id roles $a   blockIdx roles $b"
done
