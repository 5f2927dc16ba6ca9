# value vs value_pipelined, unordered two-stream mode and priority-stream mode
for c in ${CODES:-27 47 29 615}; do
  for m in "" --pipeline-priority; do timeout -k 10 200 python bench.py --code $c --no-cpu-baseline --steps 40 $m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${m:-two-streams}', d['config']['workload'][:12], d['value'], d['value_pipelined'])"; done; done
