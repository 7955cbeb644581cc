L=$PWD/topdogspectrumanalyser_amd
TDSA_HIP_LIB=$L/libtdsa_emit.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c5 or long or welch" 2>&1 | tail -2
bash tools/c5_ab.sh 2 hip emit 2>&1 | tail -3
