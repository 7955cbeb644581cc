#!/bin/bash
# Collects the round's committed evidence on the GPU box (run through gpurun from the repo root):
# bench lines for every config, rocprofv3 kernel stats of the same bench command, HBM counters (separate passes).
set -x
OUT=gpurun_out/prof_round
mkdir -p $OUT && export TMPDIR=/tmp
python bench.py > $OUT/bench_c3.json 2> $OUT/bench.err
for c in c2 c4 c5; do python bench.py --config $c --steps 300 --warmup 50 > $OUT/bench_$c.json 2>> $OUT/bench.err; done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --streams 1 --steps 300 --warmup 50 --no-cpu-baseline > $OUT/stats_run.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_rd -- python tools/devbench.py --steps 3 --warmup 1 --hold 1 > $OUT/pmc_rd.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_wr -- python tools/devbench.py --steps 3 --warmup 1 --hold 1 > $OUT/pmc_wr.log 2>&1
python tools/pipebench.py > $OUT/pipebench.txt 2>&1
python tools/analyticsbench.py > $OUT/analyticsbench.txt 2>&1
python tools/overlapbench.py > $OUT/overlapbench.txt 2>&1
find $OUT -name "*.csv" | head -20
