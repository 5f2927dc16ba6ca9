# pipelined (two handles, priority streams) vs serial decode rate for the register-kernel codes
for c in ${CODES:-27 47 29 49 615 spiral27}; do timeout -k 10 200 python bench.py --code $c --pipeline --no-cpu-baseline --steps 20 > gpurun_out/pipe_$c.json 2>gpurun_out/pipe_$c.err && python -c "
import json,sys
d=json.loads(open('gpurun_out/pipe_$c.json').read().strip().splitlines()[-1])
print('$c', 'serial', d['value'], 'pipelined', d['value_pipelined'], 'update_ms', d['update_ms'], 'chainback_ms', d['chainback_ms'])
" || exit 1; done
