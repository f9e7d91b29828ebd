#!/bin/bash
# Proposal-kernel time against the tile shape (sx cells by ry rows of one layer), forced for every density level through
# WOL_FORCE_ROW_TILE: what the tile search of build_tile_tables (wol_api.hip) should be choosing.
#   bash tools/tile_sweep.sh "<sx list>" "<ry list>" [bench flags...]      e.g.  bash tools/tile_sweep.sh "34 17 12" "3 4 5 6 7" --config 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
SXS=$1; RYS=$2; shift 2
run() {
    python3 $R/bench.py --no-cpu-baseline --no-end-to-end --steps 6 --warmup 2 "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('%-9s step %.3f ms  proposal kernel %.3f ms  exact-path share %.5f  level %s  tile %s of grid %s' % ('$TAG', d['ms_per_step'], r['avg_launch_ms'], r['exact_path_fraction_last_launch'], r['tile_density_level'], r['tile_cells'], r['grid_cells']))"
}
TAG=default run "$@"
for sx in $SXS; do for ry in $RYS; do
    TAG="${sx}x${ry}" WOL_FORCE_ROW_TILE=$sx,$ry run "$@"
done; done
TAG=default run "$@"
