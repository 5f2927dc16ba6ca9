mkdir -p gpurun_out/r2r
for c in 27 47 spiral27 29 49 spiral29; do for fr in 0 131072 8192; do for lb in 0 1 2; do
  if [ $fr = 0 ]; then FA=""; else FA="--frames $fr"; fi
  timeout -k 10 120 python bench.py --code $c $FA --variant $((2 | ((lb + 1) << 8))) --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$c', '$fr', $lb, d['config']['chunk_frames'], round(d['value']), d['ms_per_step'], d['update_ms'], d['chainback_ms'], d['bit_errors'], d['roofline']['alone']['update_ms'], d['roofline']['alone']['chainback_ms'])"
done; done; done > gpurun_out/r2r/lbsweep.txt 2>&1
cat gpurun_out/r2r/lbsweep.txt
