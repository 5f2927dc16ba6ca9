#!/bin/bash
# Collect the rocprofv3 evidence kept under profiles/ (run on the GPU box from the repo root):
#   kernel-trace stats of the default bench and of the K=15 / K=24 configs, VALU-instruction and HBM-traffic PMC passes.
# Counter passes run on their own (no trace domains besides --kernel-trace), one counter set per run.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in 27 615 224; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_$c -o s --output-format csv -- python3 $R/bench.py --code $c --no-cpu-baseline --no-pipeline --steps 5 --warmup 1 > $O/stats_$c.log 2>&1
  echo "stats $c done"
done
for c in 27 615; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE -d $O/valu_$c -o v --output-format csv -- python3 $R/bench.py --code $c --no-cpu-baseline --no-pipeline --steps 2 --warmup 1 > $O/valu_$c.log 2>&1
  echo "valu $c done"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch_$c -o f --output-format csv -- python3 $R/bench.py --code $c --no-cpu-baseline --no-pipeline --steps 2 --warmup 1 > $O/fetch_$c.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write_$c -o w --output-format csv -- python3 $R/bench.py --code $c --no-cpu-baseline --no-pipeline --steps 2 --warmup 1 > $O/write_$c.log 2>&1
  echo "traffic $c done"
done
ls -R $O | head -60
