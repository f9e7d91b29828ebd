#!/bin/bash
# Runs on the GPU box (via gpurun): per-kernel time and SQ / memory counters of the default bench workload
# (256 frames x 100k waters per step).  rocprofv3 gets the program itself after `--`; counters are collected in their own
# passes (8 SQ slots per pass; FETCH_SIZE and WRITE_SIZE cannot share one).  Output: gpurun_out/prof_r02/*; summarise with
#   python tools/summarize_pmc.py gpurun_out/prof_r02 profiles/r02_sq_counters.json
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-prof_r02}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-end-to-end --steps 4 --warmup 1"
rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-end-to-end --steps 10 --warmup 2 > $O/stats.log 2>&1
echo "stats done"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_a --output-format csv -- $B > $O/pmc_a.log 2>&1
echo "pmc a done"
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/pmc_b --output-format csv -- $B > $O/pmc_b.log 2>&1
echo "pmc b done"
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch --output-format csv -- $B > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write --output-format csv -- $B > $O/pmc_write.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $O/pmc_grbm --output-format csv -- $B > $O/pmc_grbm.log 2>&1 || true
echo "pmc memory done"
# keep only what the summary needs (kernel names of torch's own kernels make the raw files large)
python3 $R/tools/summarize_pmc.py $O $O/summary.json '{"waters": 100000, "frames_per_step": 256, "dtype": "f64", "command": "bench.py --no-cpu-baseline --no-end-to-end --steps 4 --warmup 1 (stats pass: --steps 10 --warmup 2)", "units": "FETCH_SIZE / WRITE_SIZE in KiB per dispatch; SQ_* as rocprofv3 reports them (SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* in quad-cycles summed over waves / SIMDs; SQ_BUSY_CYCLES in cycles summed over the 32 shader engines; GRBM_GUI_ACTIVE in cycles summed over the 8 XCDs)"}' > $O/summary.txt
find $O -name "*counter_collection.csv" -delete
find $O -name "*kernel_trace.csv" -delete
tail -n 40 $O/summary.txt
