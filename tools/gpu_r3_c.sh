#!/bin/bash
# round 3, GPU call C: long-frame tests after the no-atomics row pass, C5 bench + per-kernel stats, group sizes, counters list
OUT=gpurun_out/r3c
rm -rf $OUT && mkdir -p $OUT && export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "long or c5 or big or welch or batched" ) > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
show() { python -c "import sys,json; d=json.loads(open('$1').read()); print('$1', d['ms_per_step'], d['parity']['pass'], d['parity']['db_err_over_allowance_x1e-3'])"; }
timeout 600 python bench.py --config c5 --no-cpu-pool --cpu-seconds 1 > $OUT/bench_c5.json 2> $OUT/bench_c5.err; show $OUT/bench_c5.json
for g in 8 16 64; do TDSA_BIG_GROUP=$g timeout 600 python bench.py --config c5 --no-cpu-baseline > $OUT/bench_c5_g$g.json 2>> $OUT/bench_c5.err; python -c "import json; d=json.load(open('$OUT/bench_c5_g$g.json')); print('group $g', d['ms_per_step'])"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c5 -- python bench.py --config c5 --steps 30 --warmup 5 --reps 3 --min-region-s 0.05 --no-cpu-baseline > $OUT/stats_c5.log 2>&1
cat $(find $OUT/stats_c5 -name "*kernel_stats.csv" | head -1) | head -8
rocprofv3 -L 2>/dev/null | grep -o "SQ_INSTS_[A-Z0-9_]*" | sort -u | tr '\n' ' ' > $OUT/counters.txt; cat $OUT/counters.txt
