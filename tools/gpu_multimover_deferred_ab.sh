#!/bin/bash
# Games of two to four movers at small / middle batches: two launches in order, deferred rollouts
# on pipe_multi_kernel (table entries through L1 / L2), and - two movers - with the pair table's
# entries staged in LDS (CAMPX_PIPE_PAIR_LDS=1).   tools/gpu_multimover_deferred_ab.sh <tag>
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
{
for g in sokoban sokoban_l1 sokoban_l2; do
  for b in 4096 8192 16384 32768; do
    echo "== $g B=$b two launches"; tools/gpu_sweep.sh $g $b 100
    echo "== $g B=$b deferred"; BENCH_FLAGS=--deferred tools/gpu_sweep.sh $g $b 100
    if [ $g = sokoban ]; then echo "== $g B=$b deferred, pair entries in LDS"; CAMPX_PIPE_PAIR_LDS=1 BENCH_FLAGS=--deferred tools/gpu_sweep.sh $g $b 100; fi
  done
done
} > $O/multi_deferred_ab.txt 2>&1
cat $O/multi_deferred_ab.txt
