#!/bin/bash
# Runs on the GPU box (via gpurun): kernel statistics and HBM-traffic counters of the default bench command.
# Usage: bash tools/collect_profiles.sh   (writes under gpurun_out/prof_*)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 3 > $R/gpurun_out/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_write --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 3 > $R/gpurun_out/prof_write.log 2>&1
cd $R && python3 bench.py > gpurun_out/bench_line.json 2> gpurun_out/bench_line.err
tail -c 600 gpurun_out/bench_line.json
