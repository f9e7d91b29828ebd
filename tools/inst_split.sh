#!/bin/bash
# Instruction counts of the proposal kernel by part: the library and its ablation builds (sweep compiled out; sweep and
# placement compiled out -- results are wrong, the counts are what is wanted) under rocprofv3 --pmc, one pass each.
#   bash tools/inst_split.sh <lib.so> [<lib.so> ...]          (bench flags through ABFLAGS)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
    O=$R/gpurun_out/inst_$$_$(basename $lib .so)
    WOL_LIB_PATH=$R/$lib rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES --kernel-trace -d $O --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-end-to-end --steps 2 --warmup 1 $ABFLAGS > /dev/null 2>&1
    python3 - "$O" "$lib" <<PY
import csv, glob, sys, collections
csv.field_size_limit(1 << 30)
acc = collections.defaultdict(float); n = collections.defaultdict(set)
for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn, newline="")):
        if "k_search_rows" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
print("%-36s" % sys.argv[2], "  ".join("%s %.4g" % (c, acc[c] / max(len(n[c]), 1)) for c in sorted(acc)), " (per launch)")
PY
    rm -rf $O
done
