timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "averag or avg or golden_batch or workgroup_chunks or reproducible" 2>&1 | tail -4
for cfg in "--nfft 8192 --hop 8192 --frames 8192" "--nfft 4096 --hop 4096 --frames 4096" "--nfft 16384 --hop 8192 --frames 2440"; do
  for old in 1 0; do
    if [ $old = 1 ]; then export TDSA_AVG_OLD=1; else unset TDSA_AVG_OLD; fi
    timeout 300 python tools/avgbench.py --avg exp 4 --steps 600 --warmup 100 $cfg 2>&1 | tail -1
  done
done
