#!/bin/bash
python tools/dcbench.py 2>&1 | tail -3
