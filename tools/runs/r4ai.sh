#!/bin/bash
mkdir -p gpurun_out/r4ai
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
O=gpurun_out/r4ai/avg.txt
for n in 256 512 1024 2048 4096; do
  f=$((20000000/n))
  python tools/avgbench.py --nfft $n --hop $n --frames $f --avg exp 4 --steps 400 --warmup 50 >> $O 2>&1
  python tools/avgbench.py --nfft $n --hop $n --frames $f --avg lin 16 --steps 400 --warmup 50 >> $O 2>&1
done
python tools/avgbench.py --avg exp 4 --steps 400 --warmup 50 >> $O 2>&1
cat $O
