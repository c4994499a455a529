#!/bin/bash
# Same-box A/B of two library builds on the default bench step (boxes differ by +-3 %, more than most single changes):
# usage: tools/ab_step.sh <libA.so> <libB.so> [rounds]   -- alternates A, B, A, B ... and prints ms/step of each run
A=$1; B=$2; n=${3:-3}
for i in $(seq 1 $n); do
  for lib in $A $B; do
    echo -n "$(basename $lib): "
    TTTS_LIB=$PWD/$lib timeout -k 10 120 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe 2>&1 | grep -o "timed steps: [0-9.]* ms/step" || exit 1
  done
done
