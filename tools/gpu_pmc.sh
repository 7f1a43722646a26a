#!/bin/bash
# Kernel trace + instruction-mix / wait PMC passes of any python script (through gpurun).
#   tools/gpu_pmc.sh <tag> <script.py> [args...]     -> gpurun_out/pmc_<tag>/
set -u
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
cmd="python3 $GRAFT_REPO_ROOT/$*"
rocprofv3 --kernel-trace --stats -d $out -o trace -- $cmd > $out/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD -d $out -o pmc_insts -- $cmd > $out/pmc_insts.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM GRBM_GUI_ACTIVE -d $out -o pmc_wait -- $cmd > $out/pmc_wait.log 2>&1
tail -5 $out/trace.log
