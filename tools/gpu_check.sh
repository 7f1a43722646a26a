#!/bin/bash
# Smoke + whole GPU test suite + the default bench line (through gpurun):  tools/gpu_check.sh <tag>
set -u
tag=$1
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$tag
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 2400 python -m pytest tests -m gpu -q --maxfail=10 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
timeout 900 python bench.py > $O/bench.log 2>$O/bench.err; echo "bench rc=$?"; tail -1 $O/bench.log | cut -c1-3000; tail -3 $O/bench.err
