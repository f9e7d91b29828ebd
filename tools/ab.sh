#!/bin/bash
# A/B of library builds inside ONE gpurun call (devices of the pool differ by several per cent):
#   bash tools/ab.sh waterorderlib_amd/libwol_base.so waterorderlib_amd/libwol.so [...]   (two alternating rounds)
# extra bench flags through ABFLAGS
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
  for lib in "$@"; do
    WOL_LIB_PATH=$R/$lib python3 $R/bench.py --no-cpu-baseline --no-end-to-end --steps 10 --warmup 3 $ABFLAGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('%-40s %8.0f frames/s  step %.3f ms  %s %.3f ms  slow %.2e  level %s tile %s' % ('$lib', d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['exact_path_fraction_last_launch'], r['tile_density_level'], r.get('tile_cells')))"
  done
done
