timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "audio" 2>&1 | tail -12 | cut -c1-220
