#!/bin/bash
# One-launch rollouts (flow: the render role follows the update role of the SAME rollout through
# published progress) against the two launches (CAMPX_NO_FLOW=1), through gpurun:
#   tools/gpu_flow_ab.sh <tag> "<batches>" "<games>" <repeats> [test]
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
if [ "${5:-}" = test ]; then
  timeout 1500 python -m pytest tests/test_fused_parity.py tests/test_tabulate.py tests/test_fuzz_parity.py tests/test_torch_ops.py tests/test_chunked_rollouts.py tests/test_deferred.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
fi
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-12s %-10s B=%7d  %8.4f ms  kernel %8.4f  frac %.3f' % ('$1', '$2', $3, d['ms_per_step'], r['kernel_ms'], r['frac']))"; }
for g in $3; do for b in $2; do for rep in $(seq 1 $4); do
  CAMPX_NO_FLOW=1 timeout 300 python bench.py --game $g --batch $b --steps 50 --warmup 20 --no-cpu-baseline --no-extras 2>>$O/stderr.log | line two-launches $g $b
  timeout 300 python bench.py --game $g --batch $b --steps 50 --warmup 20 --no-cpu-baseline --no-extras 2>>$O/stderr.log | line one-launch $g $b
done; done; done | tee $O/flow_ab.txt
