#!/bin/bash
mkdir -p gpurun_out/r4w
L=$PWD/topdogspectrumanalyser_amd
O=gpurun_out/r4w/hold.log
for rep in 1 2 3; do
  for lib in hip late h3old; do for h in 1 3; do
    [ $h = 1 ] && [ $lib != hip ] && continue
    echo "rep $rep lib=$lib hold=$h" >> $O
    TDSA_HIP_LIB=$L/libtdsa_$lib.so timeout 120 python tools/devbench.py --nfft 1024 --hop 1024 --frames 19531 --steps 3000 --warmup 500 --hold $h 2>&1 | tail -1 >> $O
  done; done
done
for h in 0 1 2 3; do
  echo "hold=$h lib=hip N=1024 (LDS table added at every hold state: regression check)" >> $O
  timeout 120 python tools/devbench.py --nfft 1024 --hop 1024 --frames 19531 --steps 3000 --warmup 500 --hold $h 2>&1 | tail -1 >> $O
done
grep -o "lib=[a-z]* hold=[0-9]\|step=[0-9.]* us" $O | paste - - | head -40
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
