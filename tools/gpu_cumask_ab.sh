#!/bin/bash
# Hiding the update pass under the previous launch's render (rollout(pipelined=True)) with the
# side stream confined to a few CUs (CAMPX_AUX_CUS = n: hipExtStreamCreateWithCUMask), against
# in-order launches and the unconfined side stream - through gpurun.
#   tools/gpu_cumask_ab.sh <tag>
set -u
tag=$1
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$tag
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_torch_ops.py -m gpu -q -x -k pipelined > $O/pytest.log 2>&1; echo "pytest (side stream unconfined) rc=$?"; tail -2 $O/pytest.log
CAMPX_AUX_CUS=16 timeout 600 python -m pytest tests/test_torch_ops.py -m gpu -q -x -k pipelined > $O/pytest16.log 2>&1; echo "pytest (16 CUs) rc=$?"; tail -2 $O/pytest16.log
for rep in 1 2; do
for g in boat_race sokoban wall_world; do
  for mode in inorder pipe pipe8 pipe16 pipe32 pipe64; do
    unset CAMPX_AUX_CUS
    flag="--pipeline"
    case $mode in
      inorder) flag="";;
      pipe) ;;
      pipe*) export CAMPX_AUX_CUS=${mode#pipe};;
    esac
    python bench.py --game $g $flag --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('CUMASK %-10s %-8s ms_per_step %.4f  kernel_ms %.4f  median %.4f  frac %.3f' % ('$g', '$mode', d['ms_per_step'], r['kernel_ms'], r['per_launch_ms']['median'], r['frac']))"
  done
done
done
