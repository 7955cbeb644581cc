L=$PWD/topdogspectrumanalyser_amd
TDSA_HIP_LIB=$L/libtdsa_perm.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c5_million or c5_reference" 2>&1 | tail -1
bash tools/c5_ab.sh 3 hip perm 2>&1 | tail -3
