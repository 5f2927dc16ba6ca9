# one line per code: decode rate (double-buffered handle), ms per step, update / chainback ms in the pipeline and alone, bit errors
# (CODES="27 47 ..." to choose)
for c in ${CODES:-27 47 29}; do timeout -k 10 200 python bench.py --code $c --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=d['roofline']['alone']
print(d['config']['workload'][:22], d['value'], d['ms_per_step'], d['update_ms'], d['chainback_ms'], a['update_ms'], a['chainback_ms'], d['bit_errors'])"; done
