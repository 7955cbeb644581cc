#!/bin/bash
mkdir -p gpurun_out/r4ah
python -m pytest tests -m gpu -x -q -k "averag" 2>&1 | tail -5
O=gpurun_out/r4ah/avg.txt
for n in 512 1024 2048; do
  f=$((20000000/n))
  for old in 1 0; do
    if [ $old = 1 ]; then export TDSA_AVG_OLD=1; else unset TDSA_AVG_OLD; fi
    python tools/avgbench.py --nfft $n --hop $n --frames $f --avg exp 4 --steps 400 --warmup 50 >> $O 2>&1
    python tools/avgbench.py --nfft $n --hop $n --frames $f --avg lin 16 --steps 400 --warmup 50 >> $O 2>&1
  done
done
cat $O
