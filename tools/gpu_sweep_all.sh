#!/bin/bash
# The round's sweep (through gpurun): fraction of peak across batch sizes - odd ones and sizes
# whose frames are not window-aligned included - and episode lengths.  tools/gpu_sweep_all.sh <tag> [wide]
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$1
if [ "${2:-}" = wide ]; then
  {
  echo "# wide tier (state-table games, boards above 128 cells): same command"
  bash tools/gpu_sweep.sh maze16 "1000 4096 16384 65535 65536 100001 262144" "100"
  bash tools/gpu_sweep.sh maze16 "65536" "16 400"
  bash tools/gpu_sweep.sh maze32 "1000 4096 16383 16384 65536" "100"
  } | tee gpurun_out/$1/sweep_wide.txt
  exit 0
fi
{
echo "# tools/gpu_sweep.sh: python bench.py --game G --batch B --frames T --steps 30 --warmup 20 (+ >= 50 ms of settle launches), one MI355X;"
echo "# frac = algorithmic bytes per env-step x B x T / HIP-event time per launch / 8 TB/s"
bash tools/gpu_sweep.sh boat_race "1000 4096 16384 65535 65536 100000 100001 200000 499984 524288" "100"
bash tools/gpu_sweep.sh boat_race "65536" "16 64 400 1000 4000"
bash tools/gpu_sweep.sh wall_world "1000 16384 65536 262143 262144 524288" "100"
bash tools/gpu_sweep.sh sokoban "1000 4096 8192 16384 65536 99999 131071 131072 262144 524288" "100"
bash tools/gpu_sweep.sh sokoban_l1 "4096 8192" "100"
bash tools/gpu_sweep.sh sokoban_l2 "8192 16384 131072" "100"
bash tools/gpu_sweep.sh hello_world "4096 16384 32768 65536" "100"
} | tee gpurun_out/$1/sweep.txt
