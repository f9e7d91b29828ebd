#!/bin/bash
# cell-width sweep of the bench workload inside ONE gpurun call: bash tools/cw_sweep.sh [lib.so] ; widths through CWS
R=${GRAFT_REPO_ROOT:-$(pwd)}
LIB=${1:-waterorderlib_amd/libwol.so}
for cw in ${CWS:-3.6 3.9 4.2 4.5 4.8}; do
  WOL_LIB_PATH=$R/$LIB python3 $R/bench.py --no-cpu-baseline --no-end-to-end --steps 10 --warmup 3 --cell-width $cw $ABFLAGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('cw %-5s %8.0f frames/s  step %.3f ms  %s %.3f ms  slow %.2e  block %s' % ('$cw', d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['exact_path_fraction_last_launch'], r['proposal_block_threads']))"
done
