#!/bin/bash
# round 3, GPU call A: full GPU suite, then the submission modes of bench.py side by side (same box)
OUT=gpurun_out/r3a
rm -rf $OUT && mkdir -p $OUT && export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
for spec in ; do
  set -- $spec
  timeout 300 python tools/devbench.py --steps 6000 --warmup 1500 --streams $1 --batch $2 >> $OUT/devbench.txt 2>&1
done
cat $OUT/devbench.txt
timeout 600 python bench.py > $OUT/bench_c3.json 2> $OUT/bench_c3.err
cat $OUT/bench_c3.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('value','value_streams','value_serial','ms_per_step','ms_per_step_streams','ms_per_step_serial')}); print(d['roofline']['frac'], d['roofline']['kernel_avg_us'], d['roofline']['single_step_launch']); print(d['parity']); print(d.get('cpu_baseline_pool'))"
tail -3 $OUT/bench_c3.err
