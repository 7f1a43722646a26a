#!/bin/bash
# Odd batch sizes: parity tests, then the sweep's odd sizes with the per-frame streams padded
# to a 16-element row pitch (FusedGame's default, CampxOutputs.scalar_pitch) and, for the A/B,
# unpadded (CAMPX_ROW_PITCH=0: rows back to back, misaligned stores) - through gpurun.
set -u
tag=$1
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$tag
mkdir -p $O
export TMPDIR=/tmp
[ -n "${SKIP_TESTS:-}" ] || { timeout 1500 python -m pytest tests/test_update_workgroups.py tests/test_fuzz_parity.py tests/test_chunked_rollouts.py tests/test_fused_parity.py tests/test_torch_ops.py tests/test_tabulate.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log; }
for mode in padded unpadded; do
  unset CAMPX_ROW_PITCH
  [ $mode = unpadded ] && export CAMPX_ROW_PITCH=0
  echo "== $mode"
  bash tools/gpu_sweep.sh boat_race "65535 65536 100001 100003" "100"
  bash tools/gpu_sweep.sh sokoban "99999 131071" "100"
  bash tools/gpu_sweep.sh wall_world "262143" "100"
done
