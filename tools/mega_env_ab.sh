#!/bin/bash
# A/B of the persistent decode engine's step time between two environment settings, alternating pairs of fresh processes
# (tools/mega_prof.py, stamps off).  Usage: tools/mega_env_ab.sh "GSV_MEGA_HINT=49679" "GSV_MEGA_HINT=49743" [pairs]
export PROF_OFF=1
A="$1"; B="$2"; N=${3:-6}
for i in $(seq 1 $N); do
  a=$(env $A timeout -k 10 100 python3 tools/mega_prof.py 2>/dev/null | grep mode | sed 's/.*= \([0-9.]*\) us\/step.*/\1/') || exit 1
  b=$(env $B timeout -k 10 100 python3 tools/mega_prof.py 2>/dev/null | grep mode | sed 's/.*= \([0-9.]*\) us\/step.*/\1/') || exit 1
  echo "$A: $a   $B: $b"
done
