#!/bin/bash
# Tuple-table (three / four movers) check: parity tests for the sokoban levels + fuzz, then
# per-kernel timings of the variants given.   tools/gpu_tuple.sh <tag> variants...
set -u
tag=$1; shift
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$tag
mkdir -p $O
timeout 1500 python -m pytest tests/test_fused_parity.py tests/test_fuzz_parity.py tests/test_torch_ops.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
tools/gpu_ktrace.sh $tag "sokoban_l1 sokoban_l2" "$@"
