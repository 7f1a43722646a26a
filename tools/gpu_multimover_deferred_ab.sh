#!/bin/bash
# Games of two to four movers at small / middle batches, four arms per size:
#   in order      two launches per rollout (bench.py as it is)
#   shared        deferred rollouts, always the shared launch (pipe_multi_kernel; CAMPX_PIPE_MULTI_MAX_B=65536,
#                 CAMPX_NO_PIPELINE_DEFERRED=1)
#   two streams   rollout(pipelined=True): the update pass on the high-priority side stream (bench.py --pipeline)
#   as shipped    deferred rollouts as rollout_deferred() routes them
#   tools/gpu_multimover_deferred_ab.sh <tag>
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
{
for g in sokoban sokoban_l1 sokoban_l2; do
  for b in 4096 8192 16384 32768 65536; do
    echo "== $g B=$b in order"; tools/gpu_sweep.sh $g $b 100
    echo "== $g B=$b shared launch"; CAMPX_PIPE_MULTI_MAX_B=65536 CAMPX_NO_PIPELINE_DEFERRED=1 BENCH_FLAGS=--deferred tools/gpu_sweep.sh $g $b 100
    echo "== $g B=$b two streams"; BENCH_FLAGS=--pipeline tools/gpu_sweep.sh $g $b 100
    echo "== $g B=$b as shipped"; BENCH_FLAGS=--deferred tools/gpu_sweep.sh $g $b 100
  done
done
} 2>&1 | grep -v "^$" | paste - - | sed 's/SWEEP [a-z_0-9]* *B= *[0-9]* T= *100 *//; s/  update_.*//; s/  pipe_.*//' > $O/multi_deferred_ab.txt
cat $O/multi_deferred_ab.txt
