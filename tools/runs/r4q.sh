L=$PWD/topdogspectrumanalyser_amd
TDSA_HIP_LIB=$L/libtdsa_rawsp.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "hackrf_plain_int8_all_sizes or c3_full or batched_captures_full" 2>&1 | tail -2
for rep in 1 2 3; do for lib in hip rawsp; do
  TDSA_HIP_LIB=$L/libtdsa_$lib.so python tools/devbench.py --steps 6000 --warmup 1504 --batch 8 2>&1 | tail -1 | cut -c1-140
  TDSA_HIP_LIB=$L/libtdsa_$lib.so python tools/devbench.py --steps 6000 --warmup 1500 2>&1 | tail -1 | cut -c1-140
done; done
