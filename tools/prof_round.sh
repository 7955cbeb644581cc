set -x
mkdir -p gpurun_out/prof2 && export TMPDIR=/tmp
python bench.py > gpurun_out/prof2/bench_c3.json 2> gpurun_out/prof2/bench_c3.err
for c in c2 c4 c5; do python bench.py --config $c --steps 50 --warmup 5 > gpurun_out/prof2/bench_$c.json 2>> gpurun_out/prof2/bench_c3.err; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof2/stats -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/prof2/stats_run.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof2/pmc_rd -- python tools/devbench.py --steps 3 --warmup 1 --hold 1 > gpurun_out/prof2/pmc_rd.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof2/pmc_wr -- python tools/devbench.py --steps 3 --warmup 1 --hold 1 > gpurun_out/prof2/pmc_wr.log 2>&1
find gpurun_out/prof2 -name "*.csv" | head -20
