#!/bin/bash
python tools/dcbench.py 2>&1 | tail -2
python -m pytest tests -m gpu -x -q -k "dc or tracked or golden or sweep or sequence" 2>&1 | tail -3
