#!/bin/bash
# LDS bank-conflict counters of the proposal kernel for one library build: bash tools/lds_conf.sh <lib.so>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/ldsc_$$
cd /tmp && export TMPDIR=/tmp
WOL_LIB_PATH=$R/$1 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU -d $O --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-end-to-end --steps 3 --warmup 1 > /dev/null 2>&1
python3 - "$O" "$1" <<PY
import csv,glob,sys,collections
csv.field_size_limit(1<<30)
acc=collections.defaultdict(float); n=collections.defaultdict(set)
for fn in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "k_search_rows" in r["Kernel_Name"]:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
print(sys.argv[2], {k: "%.4g" % (v/len(n[k])) for k,v in acc.items()})
PY
rm -rf $O
