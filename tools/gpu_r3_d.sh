#!/bin/bash
OUT=gpurun_out/r3d
rm -rf $OUT && mkdir -p $OUT && export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "long or c5 or big or welch" ) > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c5 -- python bench.py --config c5 --steps 30 --warmup 5 --reps 3 --min-region-s 0.05 --no-cpu-baseline > $OUT/stats_c5.log 2>&1
cat $(find $OUT/stats_c5 -name "*kernel_stats.csv" | head -1) | head -5
for g in 32 64; do TDSA_BIG_GROUP=$g timeout 600 python bench.py --config c5 --no-cpu-baseline > $OUT/bench_c5_g$g.json 2>> $OUT/bench_c5.err; python -c "import json; d=json.load(open('$OUT/bench_c5_g$g.json')); print('group $g', d['ms_per_step'])"; done
for g in 32 64; do TDSA_BIG_GROUP=$g timeout 600 python bench.py --config c5 --no-cpu-baseline > $OUT/bench_c5_g$g.json 2>> $OUT/bench_c5.err; python -c "import json; d=json.load(open('$OUT/bench_c5_g$g.json')); print('group $g', d['ms_per_step'])"; done
