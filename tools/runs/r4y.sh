#!/bin/bash
mkdir -p gpurun_out/r4y
python tools/c5_scaling.py > gpurun_out/r4y/scaling.txt 2>&1
cat gpurun_out/r4y/scaling.txt
