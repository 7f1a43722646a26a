#!/bin/bash
# Overlapped small-batch rollouts (one persistent launch) against the two-launch path
# (through gpurun):  tools/gpu_overlap_ab.sh <tag> "<batches>" "<games>" [test]
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
if [ "${4:-}" = test ]; then
  timeout 1500 python -m pytest tests/test_fused_parity.py tests/test_tabulate.py tests/test_fuzz_parity.py tests/test_torch_ops.py tests/test_chunked_rollouts.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
fi
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-10s %-10s B=%7d  %8.4f ms  kernel %8.4f  frac %.3f' % ('$1', '$2', $3, d['ms_per_step'], r['kernel_ms'], r['frac']))"; }
for g in $3; do for b in $2; do
  CAMPX_OVERLAP=0 python bench.py --game $g --batch $b --steps 30 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | line two-launch $g $b
  CAMPX_OVERLAP_MAX_B=1000000 python bench.py --game $g --batch $b --steps 30 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | line overlapped $g $b
done; done | tee $O/overlap_ab.txt
