cd ${GRAFT_REPO_ROOT:-.}
for E in 0 1; do
  echo "== SK_MANY_TWO_STREAMS=$E"
  SK_MANY_TWO_STREAMS=$E timeout -k 10 600 python3 bench.py --pairs 1000000 --steps 2 --warmup 1 --cpu-sample 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for x in d['extra']['rates']:
    if 'frac_many' in x and 'trim' in x['config']: print(x['config'][:50], 'per call', x['frac'], 'pipelined', x['frac_pipelined'], 'many', x['frac_many'], x['ms_many'])
"
done
