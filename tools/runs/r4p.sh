timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
bash tools/c5_ab.sh 2 hip plain sc1 2>&1 | tail -4
python bench.py --config c5 --cpu-seconds 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5 step', round(d['ms_per_step']*1e3,1), 'frac', round(d['roofline']['frac'],4), d['parity']['pass'])"
