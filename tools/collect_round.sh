#!/bin/bash
# Per-round evidence run on the GPU box: bench lines of the three workloads, rocprofv3 kernel stats of the default bench
# command, and the two --pmc passes (FETCH_SIZE / WRITE_SIZE) the HBM-traffic table is built from.  Everything lands in
# gpurun_out/r02/; the files to keep are then copied into profiles/.   usage: bash tools/collect_round.sh <tag>
set -o pipefail
tag=${1:-r05}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py --steps 20 --warmup 5 > $out/bench_base_b64.json 2> $out/bench_base_b64.err || exit 1
echo "base b64 done"
python3 bench.py --steps 20 --warmup 5 --batch 16 --no-cpu-baseline > $out/bench_base_b16.json 2> $out/bench_base_b16.err || exit 1
echo "base b16 done"
python3 bench.py --steps 10 --warmup 4 --config scaled --batch 32 --no-cpu-baseline > $out/bench_scaled_b32.json 2> $out/bench_scaled_b32.err || exit 1
echo "scaled b32 done"
python3 bench.py --steps 20 --warmup 5 --ragged --no-cpu-baseline > $out/bench_base_b64_ragged.json 2> $out/bench_base_b64_ragged.err || exit 1
echo "ragged done"
python3 bench.py --steps 32 --warmup 16 --ragged --cycle 16 --no-cpu-baseline > $out/bench_base_b64_cycle16.json 2> $out/bench_base_b64_cycle16.err || exit 1
echo "cycle16 (16 distinct ragged batches, shape-keyed graph cache) done"
python3 bench.py --steps 16 --warmup 4 --batch 16 --accumulate 4 --no-cpu-baseline > $out/bench_base_b16_acc4.json 2> $out/bench_base_b16_acc4.err || exit 1
echo "accumulate 4 x 16 done"
MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python3 tools/rccl_one_rank.py > $out/rccl_one_rank.json 2> $out/rccl_one_rank.err || exit 1
echo "rccl one-rank done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 8 --warmup 3 --sustain 3 --no-cpu-baseline --no-probe > $out/stats.log 2>&1 || exit 1
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/traf_fetch -- python3 bench.py --steps 3 --warmup 3 --sustain 0 --no-cpu-baseline --no-probe > $out/traf_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/traf_write -- python3 bench.py --steps 3 --warmup 3 --sustain 0 --no-cpu-baseline --no-probe > $out/traf_write.log 2>&1 || exit 1
python3 tools/collect_traffic.py $out/traf_fetch $out/traf_write $out/traffic.json "$tag build, bench.py --steps 3 --warmup 3 --sustain 0, separate --pmc FETCH_SIZE / WRITE_SIZE passes" > /dev/null || exit 1
cp $out/stats/*/*kernel_stats.csv $out/kernel_stats.csv
rm -rf $out/traf_fetch $out/traf_write
echo "traffic done"
