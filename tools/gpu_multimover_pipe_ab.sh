#!/bin/bash
# Multi-mover games at small / middle batches: two launches in order against the update pass of
# rollout i+1 on a side stream under the render of rollout i (bench.py --pipeline), side stream at
# normal and high priority.   tools/gpu_multimover_pipe_ab.sh <tag>
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
{
for g in sokoban sokoban_l2; do
  for b in 4096 8192 16384 32768; do
    echo "== $g B=$b in-order"; tools/gpu_sweep.sh $g $b 100
    echo "== $g B=$b --pipeline"; BENCH_FLAGS=--pipeline tools/gpu_sweep.sh $g $b 100
    echo "== $g B=$b --pipeline, high-priority side stream"; CAMPX_AUX_PRIORITY=-1 BENCH_FLAGS=--pipeline tools/gpu_sweep.sh $g $b 100
  done
done
} > $O/pipe_ab.txt 2>&1
cat $O/pipe_ab.txt
echo "== headline, 20 driver-style runs, RCCL stream at high priority (default) / normal"
timeout 1500 python tools/driver_repro.py 20 > $O/repro_high.txt 2>&1; cat $O/repro_high.txt
timeout 900 python tools/driver_repro.py 10 -- --pg-priority normal > $O/repro_normal.txt 2>&1; cat $O/repro_normal.txt
