#!/bin/bash
mkdir -p gpurun_out/r4ag
O=gpurun_out/r4ag/avg.txt
for n in 512 1024 2048 4096; do
  f=$((20000000/n))
  python tools/avgbench.py --nfft $n --hop $n --frames $f --avg exp 4 --steps 400 --warmup 50 >> $O 2>&1
  python tools/devbench.py --nfft $n --hop $n --frames $f --steps 400 --warmup 50 --hold 0 --mode pow 2>&1 | tail -1 >> $O
done
cat $O
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4ag/prof -- python $GRAFT_REPO_ROOT/tools/avgbench.py --nfft 1024 --hop 1024 --frames 19531 --avg exp 4 --steps 300 --warmup 50 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; head -8 gpurun_out/r4ag/prof/*/*kernel_stats.csv | cut -c1-200
