#!/bin/bash
# Same-box A/B of one bench.py switch on the default step: alternates `bench.py` and `bench.py <flag>`.  usage: tools/ab_flag.sh --no-head-images [rounds]
flag=$1; n=${2:-3}
for i in $(seq 1 $n); do
  for f in "" "$flag"; do
    echo -n "bench.py $f: "
    timeout -k 10 150 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe --sustain 0 --no-alignments-figure $f 2>&1 | grep -o "timed steps: [0-9.]* ms/step" || exit 1
  done
done
