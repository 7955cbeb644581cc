#!/bin/bash
# Round 6 evidence on the GPU box (run through gpurun from the repo root); summaries are copied into profiles/r06_* by hand.
set -x
OUT=gpurun_out/prof_r06
rm -rf $OUT && mkdir -p $OUT && export TMPDIR=/tmp
( time python bench.py > $OUT/bench_c3.json 2> $OUT/bench.err ) 2> $OUT/bench_c3_time.txt
# eight ranks on this ONE GPU: plumbing and parity of the N = 8 launch (no rate: the ranks time-slice one device)
python bench.py --gpus 8 --reps 2 --min-region-s 0.1 --no-cpu-baseline --no-other-configs > $OUT/bench_c3_8ranks.json 2>> $OUT/bench.err
python bench.py --gpus 8 --config c4 --reps 2 --min-region-s 0.1 --no-cpu-baseline > $OUT/bench_c4_8ranks.json 2>> $OUT/bench.err
python bench.py --gpus 8 --config c5 --reps 2 --min-region-s 0.1 --no-cpu-baseline > $OUT/bench_c5_8ranks.json 2>> $OUT/bench.err
FAST="--reps 3 --min-region-s 0.15 --no-cpu-baseline --no-parity --no-other-configs --no-frame-stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c3_batch -- python bench.py --legs value --streams 1 --batch 8 --steps 800 --warmup 100 $FAST > $OUT/stats_c3_batch.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c5 -- python bench.py --config c5 --steps 30 --warmup 5 --reps 3 --min-region-s 0.05 --no-cpu-baseline --no-parity > $OUT/stats_c5.log 2>&1
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -size +20M -delete
