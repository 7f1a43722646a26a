#!/bin/bash
# deferred rollouts around the bounds of the shared launch (odd batch, 65 536 / 65 600 / 200 000 environments, wall world
# above 2 GB per rollout, a two-mover game, T = 1 000): tests, then bench.py --deferred sweeps (through gpurun).  The spreading
# experiment of profiles/r04_deferred_ab.txt section 5 ran from this file with CAMPX_PIPE_STRIDE / CAMPX_PIPE_SPREAD_FROM, since removed.
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_deferred.py -m gpu -x -q 2>&1 | tail -4
export BENCH_FLAGS=--deferred
timeout 400 bash tools/gpu_sweep.sh boat_race "1000 65536 65600 200000" "100"
timeout 400 bash tools/gpu_sweep.sh wall_world "32768 65536" "100"
timeout 400 bash tools/gpu_sweep.sh sokoban "16384" "100"
timeout 400 bash tools/gpu_sweep.sh boat_race "16384" "1000"
