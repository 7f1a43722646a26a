#!/bin/bash
# round 4: Engine.play() - parity, per-call timings at B = 65 536 / 131 072 (new and round-3
# kernels), and the kernel trace that goes to profiles/r04_play_rocprofv3.txt
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r4playf}
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_fused_parity.py tests/test_tabulate.py tests/test_tabulate_hidden.py tests/test_fuzz_parity.py tests/test_torch_ops.py tests/test_wide_parity.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for knobs in none CAMPX_NO_ROWS_STEP=1 none CAMPX_NO_ROWS_STEP=1; do
  echo "== per call, $knobs"
  ( [ "$knobs" != none ] && export "$knobs"; timeout 600 python tools/bench_play_games.py 2>&1 | grep "B= 65536\|B=131072" ) | tee -a $O/per_call_$knobs.txt
done
bash tools/gpu_play_games_ab.sh $(basename $O)/trace 65536 none CAMPX_NO_ROWS_STEP=1 CAMPX_STEP_LDS_TABLE=1 > $O/trace_65536.txt 2>&1
bash tools/gpu_play_games_ab.sh $(basename $O)/trace 131072 none CAMPX_NO_ROWS_STEP=1 > $O/trace_131072.txt 2>&1
cat $O/trace_65536.txt $O/trace_131072.txt | grep "==\|step_\|Fill" | awk '/==|step_/ || / 100 /' | cut -c1-150
