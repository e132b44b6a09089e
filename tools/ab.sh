#!/bin/bash
# A/B of two builds of liblce_hip.so on the same box, interleaved so that clock drift hits both.
# usage: bash tools/ab.sh <exp.so> <rounds> -- <run_one.py args>...   (several arg sets separated by ';;')
EXP=$1; ROUNDS=$2; shift 3
IFS=';' read -ra SETS <<< "$(echo "$@" | sed 's/;;/;/g')"
for r in $(seq 1 $ROUNDS); do
  for s in "${SETS[@]}"; do
    [ -z "$s" ] && continue
    a=$(python tools/run_one.py $s 2>/dev/null | tail -1)
    b=$(LCE_HIP_LIBRARY=$EXP python tools/run_one.py $s 2>/dev/null | tail -1)
    echo "base $a | exp $b"
  done
done
