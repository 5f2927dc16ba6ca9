#!/bin/bash
# Collect the evidence kept under profiles/ (run on the GPU box from the repo root):  tools/collect_profiles.sh r03 [part]
# part: "pmc" (kernel stats + counter passes), "sweep" (per-code sweep, shard-size lines, harness, probes) or empty for both -- the
# two halves fit one gpurun call each
#   1. rocprofv3 --kernel-trace --stats of the default bench command (what the driver runs) and of the serial schedule,
#      the K=15 and the K=24 configs
#   2. PMC passes, each on its own (MI355X_MICROARCH.md §HBM / §PMC slots: FETCH_SIZE and WRITE_SIZE do not fit one pass):
#      SQ_INSTS_VALU..., FETCH_SIZE, WRITE_SIZE for K=7 / K=15 / K=24; summarised into profiles/{traffic,valu}_<code>.json
#      together with a fingerprint of the kernel sources (tools/kernel_hash.py) so that bench.py can tell when they are stale
#   3. the per-code sweep with the CPU baseline, the config-5 shard-size lines, the harness in its default and in its
#      reference-methodology (--host-api) mode
# The program goes directly after `--` (no env / bash -c hop).
TAG=${1:-r03}
PART=${2:-all}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
P=$O/profiles   # gpurun merges only gpurun_out/ back: copy $P/* into profiles/ afterwards
export PROFILES_DIR=$P
mkdir -p $O $P
cd /tmp && export TMPDIR=/tmp
prof() { # name, code, extra bench args...
  local name=$1 code=$2; shift 2
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_$name -o s --output-format csv -- python3 $R/bench.py --code $code --no-cpu-baseline "$@" > $O/stats_$name.json 2> $O/stats_$name.err
  cp $O/stats_$name/s_kernel_stats.csv $P/${TAG}_${name}_kernel_stats.csv 2>/dev/null
  tail -n 1 $O/stats_$name.json > $P/${TAG}_${name}_bench_under_rocprof.json
  echo "stats $name done"
}
if [ "$PART" = all ] || [ "$PART" = pmc ]; then
prof viterbi27 27 --steps 20 --warmup 3
prof viterbi27_serial 27 --steps 20 --warmup 3 --no-pipeline --no-extra-configs
prof viterbi29_serial 29 --steps 10 --warmup 2 --no-pipeline
prof viterbi49_serial 49 --steps 10 --warmup 2 --no-pipeline
prof viterbi615 615 --steps 5 --warmup 1
prof viterbispiral615 spiral615 --steps 5 --warmup 1
prof viterbi224 224 --steps 5 --warmup 1
pmc() { # name, code, counters...
  local name=$1 code=$2; shift 2
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" -d $O/pmc_$name -o c --output-format csv -- python3 $R/bench.py --code $code --no-cpu-baseline --no-extra-configs --no-pipeline --steps 2 --warmup 1 > $O/pmc_$name.log 2>&1
  echo "pmc $name done"
}
for c in 27 615 224; do
  pmc fetch_$c $c FETCH_SIZE
  pmc write_$c $c WRITE_SIZE
done
for c in 27 29 49 615 spiral615; do pmc valu_$c $c SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE; done
pmc lds_spiral615 spiral615 SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES
cd $R
python3 tools/summarize_pmc.py 27 $O/pmc_fetch_27/c_counter_collection.csv $O/pmc_write_27/c_counter_collection.csv acs_regs_kernel
python3 tools/summarize_pmc.py 615 $O/pmc_fetch_615/c_counter_collection.csv $O/pmc_write_615/c_counter_collection.csv acs_k15_kernel
python3 tools/summarize_pmc.py 224 $O/pmc_fetch_224/c_counter_collection.csv $O/pmc_write_224/c_counter_collection.csv acs_k24t_pass
# the K=15 batch chainback (speculative tree: 63 four-byte loads per 6 decoded bits) in the same passes
python3 tools/summarize_pmc.py 615 $O/pmc_fetch_615/c_counter_collection.csv $O/pmc_write_615/c_counter_collection.csv chainback_spec_kernel 615_chainback
ms27=$(python3 -c "import json;print(json.load(open('$P/${TAG}_viterbi27_serial_bench_under_rocprof.json'))['update_ms'])")
ms615=$(python3 -c "import json;print(json.load(open('$P/${TAG}_viterbi615_bench_under_rocprof.json'))['roofline']['alone']['update_ms'])")
ms29=$(python3 -c "import json;print(json.load(open('$P/${TAG}_viterbi29_serial_bench_under_rocprof.json'))['update_ms'])")
ms49=$(python3 -c "import json;print(json.load(open('$P/${TAG}_viterbi49_serial_bench_under_rocprof.json'))['update_ms'])")
mss615=$(python3 -c "import json;print(json.load(open('$P/${TAG}_viterbispiral615_bench_under_rocprof.json'))['roofline']['alone']['update_ms'])")
python3 tools/summarize_valu.py 27 $O/pmc_valu_27/c_counter_collection.csv acs_regs_kernel $ms27
python3 tools/summarize_valu.py 615 $O/pmc_valu_615/c_counter_collection.csv acs_k15_kernel $ms615
python3 tools/summarize_valu.py 29 $O/pmc_valu_29/c_counter_collection.csv acs_regs_kernel $ms29
python3 tools/summarize_valu.py 49 $O/pmc_valu_49/c_counter_collection.csv acs_regs_kernel $ms49
python3 tools/summarize_valu.py spiral615 $O/pmc_valu_spiral615/c_counter_collection.csv acs_k15_kernel $mss615
for c in 27 29 49 615 spiral615; do cp $O/pmc_valu_$c/c_counter_collection.csv $P/${TAG}_viterbi${c}_pmc_valu_counter_collection.csv; done
cp $O/pmc_lds_spiral615/c_counter_collection.csv $P/${TAG}_viterbispiral615_pmc_lds_counter_collection.csv
for c in 27 615 224; do for k in fetch write; do cp $O/pmc_${k}_$c/c_counter_collection.csv $P/${TAG}_viterbi${c}_pmc_${k}_counter_collection.csv; done; done
fi
if [ "$PART" = all ] || [ "$PART" = sweep ]; then
cd $R
# 3. sweep, shard-size lines, harness
python3 tools/sweep.py --out $P/${TAG}_sweep --steps 20 --cpu > $O/sweep.log 2>&1
echo "sweep done"
: > $P/${TAG}_config5_shard_size.jsonl
for c in 27 47 29 49; do python3 bench.py --code $c --frames 131072 --steps 5 --warmup 1 --no-cpu-baseline 2>> $O/shard.err | tail -n 1 >> $P/${TAG}_config5_shard_size.jsonl; done
python3 bench.py --code 615 --frames 131072 --steps 2 --warmup 1 --no-cpu-baseline 2>> $O/shard.err | tail -n 1 >> $P/${TAG}_config5_shard_size.jsonl
python3 bench.py --code 224 --frames 12 --payload-bits 64 --steps 2 --warmup 1 --no-cpu-baseline 2>> $O/shard.err | tail -n 1 >> $P/${TAG}_config5_shard_size.jsonl
echo "shard-size lines done"
: > $P/${TAG}_windowed_decode.jsonl
for c in 27 47 29 49; do python3 bench.py --code $c --windowed --steps 20 --warmup 3 --no-cpu-baseline 2>> $O/shard.err | tail -n 1 >> $P/${TAG}_windowed_decode.jsonl; done
for c in 615 spiral615; do python3 bench.py --code $c --windowed --steps 5 --warmup 1 --no-cpu-baseline 2>> $O/shard.err | tail -n 1 >> $P/${TAG}_windowed_decode.jsonl; done
python3 bench.py --code 615 --windowed --frames 131072 --steps 1 --warmup 1 --no-cpu-baseline 2>> $O/shard.err | tail -n 1 >> $P/${TAG}_windowed_decode.jsonl
python3 bench.py --code 27 --windowed --frames 1048576 --steps 3 --warmup 1 --no-cpu-baseline 2>> $O/shard.err | tail -n 1 >> $P/${TAG}_windowed_decode.jsonl
echo "windowed lines done"
./harness/viterbi_bench -t 1.0 -n 8 -o $P/${TAG}_harness_default.json > $O/harness.log 2>&1
./harness/viterbi_bench -t 1.0 -n 8 --host-api --hard -o $P/${TAG}_harness_host_1frame.json >> $O/harness.log 2>&1
python3 tools/tabulate_results.py $P/${TAG}_harness_default.json $P/${TAG}_harness_host_1frame.json > $P/${TAG}_harness_tables.md 2>> $O/harness.log
make -C tools > /dev/null 2>&1
./tools/valu_rate > $P/${TAG}_valu_rate.txt 2>&1
./tools/sstore_rate > $P/${TAG}_sstore_rate.txt 2>&1
./tools/icache_probe > $P/${TAG}_icache_probe.txt 2>&1
python3 tools/small_batch_probe.py > $P/${TAG}_small_batch_probe.txt 2>&1
fi
ls $P | grep $TAG
