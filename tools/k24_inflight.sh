#!/bin/bash
# K=24 with several frames per handle: how many decodes in flight (internal streams) are best?  Experiment libraries are built with
#   hipcc ... -DVH_K24_WORKERS=N -c viterbi_hip_api.hip ; hipcc -shared ... -o csrc/libviterbi_hip_wN.so   
R=${GRAFT_REPO_ROOT:-$(pwd)}
run() {
  timeout -k 10 200 python3 $R/bench.py --code 224 --frames ${2:-12} --no-cpu-baseline --no-extra-configs --steps 3 --warmup 1 2>/dev/null | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'frames', d['config']['frames_per_gpu'], 'ms_per_step', d['ms_per_step'], 'per frame', round(d['ms_per_step']/d['config']['frames_per_gpu'],4), 'frac', d['roofline']['frac'], 'errors', d['bit_errors'])"
}
run "in flight 3 (shipping)"
for n in 4 5 6; do
  L=$R/ka9q_viterbi_comparison_amd/csrc/libviterbi_hip_w$n.so
  [ -f $L ] && VITERBI_HIP_LIB=$L run "in flight $n"
done
