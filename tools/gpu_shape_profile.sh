#!/bin/bash
# Kernel trace + PMC passes of the shape tier (tools/bench_shapes.py: Hello World with trails,
# the same art without trails on the serial kernel and on the two-kernel path) - through gpurun.
#   tools/gpu_shape_profile.sh <tag>   -> gpurun_out/<tag>/summary.txt
set -u
tag=$1
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp
cmd="python3 $GRAFT_REPO_ROOT/tools/bench_shapes.py"
timeout 600 rocprofv3 --kernel-trace --stats -d $out -o trace -- $cmd > $out/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD -d $out -o pmc_insts -- $cmd > $out/pmc_insts.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM GRBM_GUI_ACTIVE -d $out -o pmc_wait -- $cmd > $out/pmc_wait.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out -o pmc_write -- $cmd > $out/pmc_write.log 2>&1
grep -v amdgpu $out/trace.log | grep "TB/s\|per launch"
cd $GRAFT_REPO_ROOT
python3 tools/rocpd_summary.py gpurun_out/$tag > gpurun_out/$tag/summary.txt 2>&1
find gpurun_out/$tag -name "*.db" -delete
head -12 gpurun_out/$tag/summary.txt | cut -c1-170
