mkdir -p gpurun_out/r4soak
( time timeout 3000 python tools/parity_soak.py --cases 3000 --long 240 --longany 240 ) > gpurun_out/r4soak/parity.txt 2>&1
( time timeout 1500 python tools/state_soak.py --trials 400 ) > gpurun_out/r4soak/state.txt 2>&1
tail -12 gpurun_out/r4soak/parity.txt; tail -5 gpurun_out/r4soak/state.txt
