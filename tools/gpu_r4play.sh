#!/bin/bash
# round 4: Engine.play() kernels - parity of everything that plays frame by frame, then
# per-game timings under the A/B knobs given as "NAME=VALUE,NAME=VALUE" arguments (one run each)
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r4play}; shift
mkdir -p $O
export TMPDIR=/tmp
if [ "${1:-}" = "test" ]; then
  shift
  timeout 1500 python -m pytest tests/test_fused_parity.py tests/test_tabulate.py tests/test_tabulate_hidden.py tests/test_fuzz_parity.py tests/test_torch_ops.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
fi
for knobs in "$@"; do
  echo "== $knobs"
  ( IFS=,; for kv in $knobs; do [ "$kv" != none ] && export "$kv"; done
    timeout 600 python tools/bench_play_games.py 2>&1 | grep "B= 65536" )
done
