#!/bin/bash
# Same-box A/B of two whole source trees (e.g. ab_old/ = an earlier commit extracted with `git archive` and built, and .):
# alternating bench runs, then rocprofv3 kernel stats of each.   usage: tools/ab_tree.sh <treeA> <treeB> <outdir> [bench args]
export TMPDIR=/tmp
A=$(realpath $1); B=$(realpath $2); out=$(realpath -m $3); shift 3
mkdir -p $out
for i in 1 2; do
  for t in $A $B; do
    echo -n "$(basename $t): "
    (cd $t && timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe "$@" 2>&1 | grep -o "timed steps: [0-9.]* ms/step") || exit 1
  done
done
for t in $A $B; do
  tag=$(basename $t)
  (cd $t && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$tag -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-probe "$@" > $out/$tag.log 2>&1) || exit 1
  cp $out/prof_$tag/*/*kernel_stats.csv $out/$tag.csv && rm -rf $out/prof_$tag
done
