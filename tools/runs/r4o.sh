L=$PWD/topdogspectrumanalyser_amd
for lib in rq1 rq2; do TDSA_HIP_LIB=$L/libtdsa_$lib.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c5_million or c5_reference" 2>&1 | tail -1; done
bash tools/c5_groups.sh "64" hip rq1 rq2 hip rq1 rq2 2>&1 | tail -8
