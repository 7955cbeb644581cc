#!/bin/bash
mkdir -p gpurun_out/r4af
python bench.py > gpurun_out/r4af/bench_c3.json 2> gpurun_out/r4af/bench_c3.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r4af/bench_c3.json').read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["parity"]["pass"], d["cpu_baseline"]["value"])
P
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
