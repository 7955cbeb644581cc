#!/bin/bash
# round 5, run 2: new bench (other_configs legs, C4 as named, C5 combine inside the timed region, both shard modes),
# Welch export / combine tests, cosine window A/B after the latency fix
set -x
OUT=gpurun_out/a2; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "welch or shader_clock or c5 or long" > $OUT/pytest.txt 2>&1; tail -5 $OUT/pytest.txt
( time python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default_time.txt; tail -3 $OUT/bench_default_time.txt; tail -5 $OUT/bench_default.err
python bench.py --config c5 --gpus 2 --cpu-seconds-multi 1 > $OUT/bench_c5_2ranks.json 2> $OUT/bench_c5_2ranks.err; tail -3 $OUT/bench_c5_2ranks.err
python bench.py --config c5 --gpus 2 --c5-shard captures --no-cpu-baseline > $OUT/bench_c5_2ranks_captures.json 2> $OUT/bench_c5_2ranks_captures.err; tail -3 $OUT/bench_c5_2ranks_captures.err
python bench.py --config c5 --gpus 2 --c5-partials f64 --no-cpu-baseline > $OUT/bench_c5_2ranks_f64.json 2>> $OUT/bench_c5_2ranks.err
python bench.py --config c4 --cpu-seconds 3 --no-cpu-pool > $OUT/bench_c4.json 2> $OUT/bench_c4.err; tail -3 $OUT/bench_c4.err
python bench.py --config c4 --gpus 2 --no-cpu-baseline > $OUT/bench_c4_2ranks.json 2> $OUT/bench_c4_2ranks.err; tail -3 $OUT/bench_c4_2ranks.err
for rep in 1 2; do
  python tools/c5_scaling.py --steps 100 --reps 1 --ks 8,16,32,64 > $OUT/scale_cos_$rep.txt 2>&1
  TDSA_HIP_LIB=$PWD/topdogspectrumanalyser_amd/libtdsa_dev.so TDSA_BIG_WIN_TABLE=1 python tools/c5_scaling.py --steps 100 --reps 1 --ks 8,16,32,64 > $OUT/scale_tab_$rep.txt 2>&1
done
grep -h "K=" $OUT/scale_*.txt
python - <<'P'
import json
for f in ("bench_default","bench_c5_2ranks","bench_c5_2ranks_captures","bench_c5_2ranks_f64","bench_c4","bench_c4_2ranks"):
    try:
        d=json.load(open(f"gpurun_out/a2/{f}.json"))
        print(f, d["value"], d["ms_per_step"], d.get("value_compute_only"), d["roofline"]["frac"], d.get("parity",{}).get("pass"), d.get("welch"), d.get("wall_s"))
        oc=d["roofline"].get("other_configs")
        if oc:
            for k,v in oc.items(): print("  ",k,{a:v.get(a) for a in ("value","ms_per_step","frac","parity","leg_wall_s","error","shader_clock_mhz")})
        print("   clock", d["roofline"].get("shader_clock_mhz"), d["roofline"].get("kernel_mcycles_per_step"))
    except Exception as e:
        print(f, "FAILED", e)
P
