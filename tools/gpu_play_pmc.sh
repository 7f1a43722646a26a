#!/bin/bash
# Counter passes of Engine.play() per game (through gpurun): tools/gpu_play_pmc.sh <tag> [B]
set -u
tag=$1; B=${2:-65536}
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp
cmd="python3 $GRAFT_REPO_ROOT/tools/play_trace_games.py $B"
timeout 600 rocprofv3 --kernel-trace --stats -d $out -o trace -- $cmd > $out/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD -d $out -o pmc_1insts -- $cmd > $out/pmc_1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $out -o pmc_2wait -- $cmd > $out/pmc_2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out -o pmc_3write -- $cmd > $out/pmc_3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out -o pmc_4fetch -- $cmd > $out/pmc_4.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/rocpd_summary.py gpurun_out/$tag > gpurun_out/$tag/summary.txt 2>&1
find gpurun_out/$tag -name "*.db" -delete
grep -c . gpurun_out/$tag/summary.txt
