#!/bin/bash
# LDS bank conflicts per kernel: rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS (MI355X_MICROARCH.md: conflict =
# extra cycles, idx_active = all LDS-array cycles).  Run on the GPU box from the repo root:  tools/lds_conflicts.sh [out-tag]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-lds}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { # name, bench args...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES -d $O/$name -o c --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > $O/$name.log 2>&1
  python3 - <<PY
import csv,collections
rows=list(csv.DictReader(open("$O/$name/c_counter_collection.csv")))
agg=collections.defaultdict(collections.Counter); n=collections.Counter()
for r in rows:
    k=r["Kernel_Name"][:70]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
for k,c in agg.items():
    if c.get("SQ_LDS_IDX_ACTIVE",0) > 0:
        m={a:b/n[(k,a)] for a,b in c.items()}
        print("$name | %-70s | conflict %.3g of %.3g LDS cycles (%.0f%%), wave cycles %.3g" % (k, m["SQ_LDS_BANK_CONFLICT"], m["SQ_LDS_IDX_ACTIVE"], 100*m["SQ_LDS_BANK_CONFLICT"]/m["SQ_LDS_IDX_ACTIVE"], m["SQ_WAVE_CYCLES"]))
PY
}
run k24 --code 224
run k15 --code 615
run k15w --code 615 --windowed
run k7 --code 27
run k7w --code 27 --windowed
run k9 --code 29
run k9w --code 29 --windowed
run k9r4 --code 49
