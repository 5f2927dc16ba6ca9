# one line per code: decode rate, update ms, chainback ms, pipelined rate (CODES="27 47 ..." to choose)
for c in ${CODES:-27 47 29}; do timeout -k 10 200 python bench.py --code $c --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['workload'][:22], d['value'], d['update_ms'], d['chainback_ms'], d['value_pipelined'], d['bit_errors'])"; done
