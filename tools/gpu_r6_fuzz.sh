#!/bin/bash
# Round 6: the seeded fuzzes at many more seeds than the suite's defaults (through gpurun).
#   tools/gpu_r6_fuzz.sh <tag>
set -u
tag=$1
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$tag
mkdir -p $O
export TMPDIR=/tmp
CAMPX_SEQ_SEEDS=${SEQ:-40} timeout 2400 python -m pytest tests/test_api_sequences.py -m gpu -q --maxfail=5 -x > $O/seq.log 2>&1; echo "sequences rc=$?"; tail -6 $O/seq.log | cut -c1-300
CAMPX_ABI_SEEDS=${ABI:-300} timeout 1500 python -m pytest tests/test_c_abi_fuzz.py -m gpu -q --maxfail=5 > $O/abi.log 2>&1; echo "abi rc=$?"; tail -4 $O/abi.log | cut -c1-300
CAMPX_FUZZ_SEEDS=${FZ:-60} CAMPX_FUZZ_TABLE_SEEDS=150 CAMPX_FUZZ_BIG_SEEDS=20 CAMPX_FUZZ_WAREHOUSE_SEEDS=20 CAMPX_FUZZ_PYTHON_SEEDS=24 timeout 2400 python -m pytest tests/test_fuzz_parity.py -m gpu -q --maxfail=5 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -4 $O/fuzz.log | cut -c1-300
