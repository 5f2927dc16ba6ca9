#!/bin/bash
# K=24 pass L: does delaying one of the two co-resident workgroups of a CU (so that its loads / stores fall into the other's
# arithmetic) shorten a lone decode?  Uses the timing build (make -C ka9q_viterbi_comparison_amd/csrc timing).
# VHIP_K24T_STAGGER = <pass mask>:<pattern>:<s_sleep units of ~64 cycles>; pattern 1: upper half of the grid, 2: odd (blockIdx>>3), 3: odd blockIdx
R=${GRAFT_REPO_ROOT:-$(pwd)}
export VITERBI_HIP_LIB=$R/ka9q_viterbi_comparison_amd/csrc/libviterbi_hip_timing.so
[ -f "$VITERBI_HIP_LIB" ] || { echo "missing $VITERBI_HIP_LIB"; exit 1; }
run() {
  VHIP_K24T_STAGGER=$1 timeout -k 10 100 python3 $R/bench.py --code 224 --no-cpu-baseline --no-extra-configs --steps 12 --warmup 3 2>/dev/null | tail -1 | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('stagger $1', 'ms_per_step', d['ms_per_step'], 'update_ms', d['update_ms'], 'errors', d['bit_errors'])"
}
run 0:0:0
run 0:0:0
for pat in 1 2 3; do for u in 8 16 24 32 48; do run 2:$pat:$u; done; done
