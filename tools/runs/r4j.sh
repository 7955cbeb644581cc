for rep in 1 2 3; do
  TDSA_BIG_SERIAL=1 python bench.py --config c5 --cpu-seconds 1 --reps 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('serial ', round(d['ms_per_step']*1e3,1), d['parity']['pass'])"
  python bench.py --config c5 --cpu-seconds 1 --reps 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('overlap', round(d['ms_per_step']*1e3,1), d['parity']['pass'])"
done
