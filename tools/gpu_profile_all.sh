#!/bin/bash
# The round's profile set (through gpurun): for every BASELINE single-GPU configuration and the
# three- / four-mover games, tools/profile.sh (kernel trace + four PMC passes of bench.py),
# summarised ON THE BOX (the rocpd databases are too large to travel back):
#   gpurun_out/<tag>/<game>_rocprofv3.txt, gpurun_out/<tag>/traffic.json
#   tools/gpu_profile_all.sh <tag> [games...]
set -u
tag=$1; shift
games=${*:-boat_race wall_world sokoban sokoban_l1 sokoban_l2}
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$tag
mkdir -p $O
specs=""
for g in $games; do
  bash tools/profile.sh ${tag}_$g --game $g > $O/profile_$g.log 2>&1
  d=gpurun_out/prof_${tag}_$g
  python3 tools/rocpd_summary.py $d > $O/${g}_rocprofv3.txt 2>&1
  b=$(python3 -c "import bench; print(bench.WORKLOADS['$g'][1])")
  specs="$specs $d:$g:$b:100"
  grep "render_kernel\|update_" $O/${g}_rocprofv3.txt | head -3 | cut -c1-140
done
python3 tools/make_traffic.py --round $(echo $tag | cut -c1-3) --out $O/traffic.json $specs > /dev/null 2>$O/traffic.err
cat $O/traffic.json | head -40
rm -rf gpurun_out/prof_${tag}_*
