#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace + PMC passes of bench.py.
#   tools/profile.sh <tag> [bench args...]
# Writes rocpd databases under gpurun_out/prof_<tag>/ ; summarise them back in the
# build container with tools/rocpd_summary.py and commit the summary to profiles/.
# Counters go in their own passes, with --kernel-trace only (see MI355X_MICROARCH.md); every pass
# under its own `timeout` (a pass that aborts inside rocprofv3 otherwise hangs until gpurun's limit).
set -u
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
bench="python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras $*"
timeout 900 rocprofv3 --kernel-trace --stats -d $out -o trace -- $bench > $out/trace.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD -d $out -o pmc_insts -- $bench > $out/pmc_insts.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $out -o pmc_wait -- $bench > $out/pmc_wait.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out -o pmc_write -- $bench > $out/pmc_write.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out -o pmc_fetch -- $bench > $out/pmc_fetch.log 2>&1
grep -h '^{' $out/trace.log | tail -1 > $out/bench_line.json
ls $out | head -20
