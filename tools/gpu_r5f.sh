#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5f; mkdir -p $O
timeout 1500 python -m pytest tests/test_tabulate_batched.py tests/test_tabulate.py tests/test_deferred.py tests/test_wide_parity.py -m gpu -x -q > $O/pytest.log 2>&1; echo rc=$?; tail -5 $O/pytest.log
{
for g in sokoban sokoban_l1 sokoban_l2; do
  for b in 16384 32768 65536; do
    echo "== $g B=$b two launches"; tools/gpu_sweep.sh $g $b 100
    echo "== $g B=$b deferred (as shipped)"; BENCH_FLAGS=--deferred tools/gpu_sweep.sh $g $b 100
  done
done
} > $O/multi_deferred_big.txt 2>&1
cat $O/multi_deferred_big.txt
