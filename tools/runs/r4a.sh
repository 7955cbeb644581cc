set -x
mkdir -p gpurun_out/r4a
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r4a/pytest.log
timeout 300 python bench.py > gpurun_out/r4a/bench_c3.json 2> gpurun_out/r4a/bench_c3.err
timeout 400 python bench.py --gpus 2 > gpurun_out/r4a/bench_c3_2r.json 2> gpurun_out/r4a/bench_c3_2r.err
timeout 400 python bench.py --gpus 2 --config c4 > gpurun_out/r4a/bench_c4_2r.json 2> gpurun_out/r4a/bench_c4_2r.err
for n in 1024 2048 4096 16384; do for h in 1 3; do
  f=$((20000000/n)); [ $n -eq 16384 ] && f=2440
  echo "N=$n hold=$h" >> gpurun_out/r4a/hold.log
  timeout 120 python tools/devbench.py --nfft $n --hop $n --frames $f --steps 3000 --warmup 500 --hold $h 2>&1 | tail -2 >> gpurun_out/r4a/hold.log
done; done
