#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5g; mkdir -p $O
timeout 900 python -m pytest tests/test_shape_parity.py tests/test_recognise.py tests/test_tabulate_batched.py -m gpu -x -q > $O/pytest.log 2>&1; echo rc=$?; tail -25 $O/pytest.log
timeout 600 python tools/bench_shapes.py > $O/bench_shapes.txt 2>&1; cat $O/bench_shapes.txt
