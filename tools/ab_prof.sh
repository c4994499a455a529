#!/bin/bash
# Same-box per-kernel A/B: rocprofv3 kernel stats of the default bench step under two library builds.
# usage: tools/ab_prof.sh <libA.so> <libB.so> <outdir>
export TMPDIR=/tmp
out=$3; mkdir -p $out
for lib in $1 $2; do
  tag=$(basename $lib .so)
  TTTS_LIB=$PWD/$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-probe > $out/$tag.log 2>&1 || exit 1
  cp $out/$tag/*/*kernel_stats.csv $out/$tag.csv && rm -rf $out/$tag
done
