#!/bin/bash
# round 3, GPU call B: bench lines of all four configurations + a sharded C5 run with two ranks on the one GPU
OUT=gpurun_out/r3b
rm -rf $OUT && mkdir -p $OUT && export TMPDIR=/tmp
show() { python -c "import sys,json; d=json.loads(open('$1').read()); print('$1', {k:d.get(k) for k in ('value','value_streams','value_serial','ms_per_step','ms_per_step_streams','ms_per_step_serial','scaling')}); print(' roofline', d['roofline']['frac'], d['roofline']['kernel_avg_us'], d['roofline'].get('single_step_launch')); print(' parity', d.get('parity')); print(' pool', d.get('cpu_baseline_pool'), d.get('welch'))"; }
timeout 600 python bench.py > $OUT/bench_c3.json 2> $OUT/bench_c3.err; show $OUT/bench_c3.json; tail -3 $OUT/bench_c3.err
timeout 600 python bench.py --batch 4 --no-cpu-baseline > $OUT/bench_c3_b4.json 2>> $OUT/bench_c3.err; show $OUT/bench_c3_b4.json
for c in c2 c4 c5; do timeout 600 python bench.py --config $c --no-cpu-pool > $OUT/bench_$c.json 2> $OUT/bench_$c.err; show $OUT/bench_$c.json; tail -3 $OUT/bench_$c.err; done
timeout 600 python bench.py --config c5 --gpus 2 > $OUT/bench_c5_2ranks.json 2> $OUT/bench_c5_2ranks.err; show $OUT/bench_c5_2ranks.json; tail -3 $OUT/bench_c5_2ranks.err
