#!/bin/bash
# counters of the shell trajectory kernels: bash tools/prof_shell.sh  (runs tools/run_shell_traj.py under rocprofv3)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_shell
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 $R/tools/run_shell_traj.py 64 > $O/stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_a --output-format csv -- python3 $R/tools/run_shell_traj.py 64 > $O/pmc_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/pmc_b --output-format csv -- python3 $R/tools/run_shell_traj.py 64 > $O/pmc_b.log 2>&1
python3 $R/tools/summarize_pmc.py $O $O/summary.json > $O/summary.txt
find $O -name "*counter_collection.csv" -delete
find $O -name "*kernel_trace.csv" -delete
grep -A18 -E "k_cos_angs_traj |k_lsi_traj |k_q_traj_points" $O/summary.txt | head -80
