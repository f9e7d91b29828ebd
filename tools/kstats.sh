#!/bin/bash
# per-kernel average times of the bench workload for one library build: bash tools/kstats.sh <lib.so>   (extra bench flags through ABFLAGS)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/kst_$$
cd /tmp && export TMPDIR=/tmp
WOL_LIB_PATH=$R/$1 rocprofv3 --kernel-trace --stats -d $O --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-end-to-end --steps 10 --warmup 2 $ABFLAGS > /dev/null 2>&1
python3 - "$O" "$1" <<PY
import csv,glob,sys
csv.field_size_limit(1<<30)
print(sys.argv[2])
for fn in glob.glob(sys.argv[1]+"/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        n=r["Name"].replace("void ","")
        if n.startswith("k_"): print("   %-40s %6s %10.1f us" % (n[:40], r["Calls"], float(r["AverageNs"])/1e3))
PY
rm -rf $O
