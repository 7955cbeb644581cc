mkdir -p gpurun_out/r4b
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r4b/pytest.log
L=$PWD/topdogspectrumanalyser_amd
for rep in 1 2; do
for n in 1024 2048 4096 8192 16384; do for h in 1 3; do
  f=$((20000000/n)); [ $n -eq 16384 ] && f=2440
  echo "N=$n hold=$h" >> gpurun_out/r4b/hold.log
  timeout 120 python tools/devbench.py --nfft $n --hop $n --frames $f --steps 3000 --warmup 500 --hold $h 2>&1 | tail -1 >> gpurun_out/r4b/hold.log
done; done
echo "N=1024 hold=3 occ3 (grid for 192 CUs x 4 = 768 workgroups, 3 per CU)" >> gpurun_out/r4b/hold.log
TDSA_NUM_CU=192 TDSA_HIP_LIB=$L/libtdsa_occ3.so timeout 120 python tools/devbench.py --nfft 1024 --hop 1024 --frames 19531 --steps 3000 --warmup 500 --hold 3 2>&1 | tail -1 >> gpurun_out/r4b/hold.log
done
# C3 shape regression check (hold=1 kernel must be untouched): batch 8
timeout 120 python tools/devbench.py --steps 6000 --warmup 1504 --batch 8 2>&1 | tail -1 >> gpurun_out/r4b/hold.log
