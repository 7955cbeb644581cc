#!/bin/bash
mkdir -p gpurun_out/r4u
python tools/c5_two_plans.py --plans 3 > gpurun_out/r4u/two_plans.txt 2>&1
cat gpurun_out/r4u/two_plans.txt
python -m pytest tests -m gpu -x -q -k "averaging" 2>&1 | tail -3
