#!/bin/bash
python -m pytest tests -m gpu -x -q -k "averag" 2>&1 | tail -3
for n in 1024 4096 16384; do
  f=$((20000000/n)); h=$n; [ $n = 16384 ] && { f=2440; h=8192; }
  python tools/avgbench.py --nfft $n --hop $h --frames $f --avg lin 100000 --steps 400 --warmup 50 2>&1 | tail -1
  python tools/avgbench.py --nfft $n --hop $h --frames $f --avg lin 100000 --steps 400 --warmup 50 --state-only 2>&1 | tail -1
done
