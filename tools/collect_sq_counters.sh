#!/bin/bash
# Runs on the GPU box (via gpurun): per-kernel time and SQ / memory counters of a bench workload.  rocprofv3 gets the
# program itself after `--`; counters are collected in their own passes (8 SQ slots per pass; FETCH_SIZE and WRITE_SIZE
# cannot share one).   bash tools/collect_sq_counters.sh <tag> [bench flags...]
#   e.g.  bash tools/collect_sq_counters.sh prof_r03                 (config 3: 256 frames x 100k waters per step)
#         bash tools/collect_sq_counters.sh prof_r03_c5 --config 5   (config 5: 32 frames x 500 100 waters, fp32)
# Output: gpurun_out/<tag>/summary.json (copy to profiles/) and the kernel_stats csv of the stats pass.
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-prof_r03}
shift || true
FLAGS="$*"
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-end-to-end --steps 4 --warmup 1 $FLAGS"
rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-end-to-end --steps 10 --warmup 2 $FLAGS > $O/stats.log 2>&1
echo "stats done"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_a --output-format csv -- $B > $O/pmc_a.log 2>&1
echo "pmc a done"
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/pmc_b --output-format csv -- $B > $O/pmc_b.log 2>&1
echo "pmc b done"
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch --output-format csv -- $B > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write --output-format csv -- $B > $O/pmc_write.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $O/pmc_grbm --output-format csv -- $B > $O/pmc_grbm.log 2>&1 || true
echo "pmc memory done"
WAT=100000; FPS=256; DT=f64
case "$FLAGS" in *"--config 5"*) WAT=500100; FPS=32; DT=f32;; esac
# keep only what the summary needs (kernel names of torch's own kernels make the raw files large)
python3 $R/tools/summarize_pmc.py $O $O/summary.json "{\"waters\": $WAT, \"frames_per_step\": $FPS, \"dtype\": \"$DT\", \"command\": \"bench.py --no-cpu-baseline --no-end-to-end --steps 4 --warmup 1 $FLAGS (stats pass: --steps 10 --warmup 2)\", \"units\": \"FETCH_SIZE / WRITE_SIZE in KiB per dispatch; SQ_* as rocprofv3 reports them (SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* in quad-cycles summed over waves / SIMDs; SQ_BUSY_CYCLES in cycles summed over the 32 shader engines; GRBM_GUI_ACTIVE in cycles summed over the 8 XCDs); vgpr = VGPR_Count of rocprofv3's kernel-trace record, which on gfx950 reads HALF the registers allocated per lane (k_search_rows: 40 for the compiler's 80; k_finish_fast: 64 for 122 used / 128 allocated; k_sort_in_row: 52 for 102 / 104): compare DESIGN's figures, which are the compiler's (.vgpr_count, -Rpass-analysis=kernel-resource-usage)\"}" > $O/summary.txt
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O -name "*counter_collection.csv" -delete
find $O -name "*kernel_trace.csv" -delete
tail -n 60 $O/summary.txt
