#!/bin/bash
# Kernel trace + PMC passes of Engine.play() (tools/play_profile.py), new and old one-frame
# kernel (through gpurun):  tools/gpu_play_profile.sh <tag>   -> gpurun_out/<tag>/{rows,old}
set -u
tag=$1
export TMPDIR=/tmp
for mode in rows old; do
  unset CAMPX_NO_ROWS_STEP
  [ $mode = old ] && export CAMPX_NO_ROWS_STEP=1
  out=$GRAFT_REPO_ROOT/gpurun_out/$tag/$mode
  mkdir -p $out
  cd /tmp
  cmd="python3 $GRAFT_REPO_ROOT/tools/play_profile.py"
  rocprofv3 --kernel-trace --stats -d $out -o trace -- $cmd > $out/trace.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD -d $out -o pmc_insts -- $cmd > $out/pmc_insts.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM GRBM_GUI_ACTIVE -d $out -o pmc_wait -- $cmd > $out/pmc_wait.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out -o pmc_write -- $cmd > $out/pmc_write.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out -o pmc_fetch -- $cmd > $out/pmc_fetch.log 2>&1
  tail -2 $out/trace.log
  cd $GRAFT_REPO_ROOT
  python3 tools/rocpd_summary.py gpurun_out/$tag/$mode > gpurun_out/$tag/summary_$mode.txt 2>&1
  head -30 gpurun_out/$tag/summary_$mode.txt
  find gpurun_out/$tag/$mode -name "*.db" -delete    # (only 64 MiB of gpurun_out travel back)
done
