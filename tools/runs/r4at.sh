#!/bin/bash
for n in 1024 4096 16384; do
  f=$((20000000/n)); h=$n; [ $n = 16384 ] && { f=2440; h=8192; }
  python tools/devbench.py --nfft $n --hop $h --frames $f --steps 600 --warmup 100 --hold 1 --fmt i8 2>&1 | tail -1
  python tools/devbench.py --nfft $n --hop $h --frames $f --steps 600 --warmup 100 --hold 1 --fmt c64 2>&1 | tail -1
done
