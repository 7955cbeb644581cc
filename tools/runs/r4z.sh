#!/bin/bash
mkdir -p gpurun_out/r4z
( time timeout 1500 python tools/parity_soak.py --cases 3000 --long 240 --longany 240 ) > gpurun_out/r4z/parity_a.txt 2>&1
( time timeout 2400 python tools/parity_soak.py --cases 6000 --long 360 --longany 360 --seed 100000 ) > gpurun_out/r4z/parity_b.txt 2>&1
( time timeout 1500 python tools/state_soak.py --trials 400 ) > gpurun_out/r4z/state.txt 2>&1
tail -9 gpurun_out/r4z/parity_a.txt; tail -9 gpurun_out/r4z/parity_b.txt; tail -5 gpurun_out/r4z/state.txt
( time TDSA_SOURCE_CASES=40 TDSA_PROCESSOR_CASES=100 TDSA_AVERAGER_CASES=100 TDSA_LONG_CASES=60 TDSA_PIPE_CASES=100 TDSA_SWEEP_CASES=400 TDSA_ANYSIZE_CASES=200 TDSA_SEQUENCE_TRIALS=100 TDSA_ANYSIZE_LONG_CASES=120 timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -3 ) > gpurun_out/r4z/pytest_big.txt 2>&1
cat gpurun_out/r4z/pytest_big.txt
