#!/bin/bash
# level-1 sort time for a library build and a batch size: bash tools/ksort.sh <lib.so> <frames per step> [frames per batch]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
WOL_LIB_PATH=$R/$1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kk --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-end-to-end --steps 6 --warmup 2 --frames-per-step $2 --frames-per-batch ${3:-0} > /dev/null 2>&1
python3 - "$1" "$2" <<PY
import csv,glob,sys
csv.field_size_limit(1<<30)
for fn in glob.glob("$R/gpurun_out/kk/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "k_sort_rows" in r["Name"] or "k_sort_in_row" in r["Name"]: print("%-28s F=%-4s %-22s %4s calls %10.1f us" % (sys.argv[1][-16:], sys.argv[2], r["Name"][5:27], r["Calls"], float(r["AverageNs"])/1e3))
PY
rm -rf $R/gpurun_out/kk
