#!/bin/bash
# Same-box A/B of the deferred second-stage reductions (ops.DEFER_REDUCE) on the default bench step: alternates on / off.
n=${1:-2}
for i in $(seq 1 $n); do
  for d in True False; do
    echo -n "DEFER_REDUCE=$d: "
    timeout -k 10 150 python3 -c "
import sys, runpy
import transformertts_amd.ops as o
o.DEFER_REDUCE = $d
sys.argv = ['bench.py', '--steps', '20', '--warmup', '5', '--no-cpu-baseline', '--no-probe']
runpy.run_path('bench.py', run_name='__main__')
" 2>&1 | grep -o "timed steps: [0-9.]* ms/step" || exit 1
  done
done
