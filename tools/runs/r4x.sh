#!/bin/bash
mkdir -p gpurun_out/r4x
L=$PWD/topdogspectrumanalyser_amd
O=gpurun_out/r4x/hold.log
for rep in 1 2 3; do
  for lib in hip h3old; do for h in 1 3; do
    [ $h = 1 ] && [ $lib != hip ] && continue
    echo "rep $rep lib=$lib hold=$h F=156248" >> $O
    TDSA_HIP_LIB=$L/libtdsa_$lib.so timeout 120 python tools/devbench.py --nfft 1024 --hop 1024 --frames 156248 --steps 600 --warmup 100 --hold $h 2>&1 | tail -1 >> $O
  done; done
done
grep -v "^$" $O | awk '/^rep|^hold=/{h=$0;next}{match($0,/step=[0-9.]+ us/); print h " -> " substr($0,RSTART,RLENGTH)}'
