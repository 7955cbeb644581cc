L=$PWD/topdogspectrumanalyser_amd
TDSA_HIP_LIB=$L/libtdsa_rowsfull.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c5 or long or welch" 2>&1 | tail -3
bash tools/c5_groups.sh "64" hip rowsfull hip rowsfull 2>&1 | tail -8
