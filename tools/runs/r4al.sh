#!/bin/bash
mkdir -p gpurun_out/r4al
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for n in 512 1024; do
f=$((20000000/n))
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4al/prof$n -- python $GRAFT_REPO_ROOT/tools/avgbench.py --nfft $n --hop $n --frames $f --avg exp 4 --steps 300 --warmup 50 > /dev/null 2>&1
head -6 $GRAFT_REPO_ROOT/gpurun_out/r4al/prof$n/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
done
