#!/bin/bash
# timeline of one bench step: kernels and copies in start order with the idle time before each (rocprofv3 kernel + memory-copy trace)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/gaps_$$
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace -d $O --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-end-to-end --steps 6 --warmup 2 $ABFLAGS > /dev/null 2>&1
python3 - "$O" <<PY
import csv,glob,sys
csv.field_size_limit(1<<30)
ev=[]
for fn in glob.glob(sys.argv[1]+"/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ","")[:44]))
for fn in glob.glob(sys.argv[1]+"/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy "+r.get("Direction","")))
ev.sort()
# last full step: from the last k_sort_rows start backwards one step
idx=[i for i,e in enumerate(ev) if e[2].startswith("k_sort_rows")]
a,b=idx[-2],idx[-1]
prev_end=ev[a-1][1] if a>0 else ev[a][0]
tot_gap=0
for s,e,n in ev[a:b]:
    gap=(s-prev_end)/1e3
    print("%9.1f us gap  %9.1f us  %s" % (gap, (e-s)/1e3, n))
    if gap>0: tot_gap+=gap
    prev_end=max(prev_end,e)
print("step %.1f us, idle %.1f us" % ((ev[b][0]-ev[a][0])/1e3, tot_gap))
PY
rm -rf $O
