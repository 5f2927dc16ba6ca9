#!/bin/bash
# Where does a K=24 tiled pass spend its time?  Kernel trace of bench.py --code 224 in the three timing modes of
# acs_k24t.hip (VHIP_K24T_MODE: 0 = real, 1 = data movement only, 2 = trellis stages + row stores only, 3 = global loads + stores only, 4 = 2 without row stores, 5 = 2 without LDS regrouping; modes 1..5
# produce wrong results by design).  Run on the GPU box from the repo root.
# The timing modes are not in libviterbi_hip.so: build the probe library first (make -C ka9q_viterbi_comparison_amd/csrc timing).
R=${GRAFT_REPO_ROOT:-$(pwd)}
export VITERBI_HIP_LIB=$R/ka9q_viterbi_comparison_amd/csrc/libviterbi_hip_timing.so
[ -f "$VITERBI_HIP_LIB" ] || { echo "missing $VITERBI_HIP_LIB (make -C ka9q_viterbi_comparison_amd/csrc timing)"; exit 1; }
O=$R/gpurun_out/${1:-k24t_probe}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in ${MODES:-0 1 2 3}; do
  VHIP_K24T_MODE=$m timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/mode$m -o s --output-format csv -- python3 $R/bench.py --code 224 --no-cpu-baseline --steps 2 --warmup 1 > $O/mode$m.log 2>&1
  echo "mode $m:"; grep -E "pass_[hl]_kernel<true" $O/mode$m/s_kernel_stats.csv | awk -F'","' '{print substr($1,1,60), "calls", $2, "avg_ns", $4, "max_ns", $7}'
done
