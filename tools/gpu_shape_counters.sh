#!/bin/bash
# Kernel trace + counter passes of shape_rollout_kernel beside render_kernel (through gpurun):
#   tools/gpu_shape_counters.sh <tag>   -> gpurun_out/<tag>/summary.txt
set -u
tag=$1
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp
cmd="python3 $GRAFT_REPO_ROOT/tools/shape_counters_workload.py"
rocprofv3 --kernel-trace --stats -d $out -o trace -- $cmd > $out/trace.log 2>&1
pass() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $out -o pmc_$name -- $cmd > $out/pmc_$name.log 2>&1 || echo "pass $name failed"; }
pass 1insts SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
pass 2wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES
pass 3active SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES
pass 4cycles SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES GRBM_GUI_ACTIVE
pass 5write WRITE_SIZE
pass 6tcc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum TCC_WRITE_sum TCC_REQ_sum TCC_BUSY_sum
pass 7tcp TCP_PENDING_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TA_BUSY_sum
cd $GRAFT_REPO_ROOT
python3 tools/rocpd_summary.py gpurun_out/$tag > gpurun_out/$tag/summary.txt 2>&1
find gpurun_out/$tag -name "*.db" -delete
grep -c . gpurun_out/$tag/summary.txt; head -8 gpurun_out/$tag/summary.txt | cut -c1-170
