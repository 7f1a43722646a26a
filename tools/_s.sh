cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_torch_ops.py tests/test_deferred.py -m gpu -x -q 2>&1 | tail -3
for g in sokoban sokoban_l1 sokoban_l2; do for b in 4096 8192 16384 32768; do
  echo "== $g B=$b deferred (as shipped)"; BENCH_FLAGS=--deferred tools/gpu_sweep.sh $g $b 100
done; done
