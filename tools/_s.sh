cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_shape_parity.py -m gpu -x -q 2>&1 | tail -3
PROBE_T="100 200" python tools/shape_update_probe.py 32768 65536 2>&1 | grep "signed char"
bash tools/gpu_sweep.sh hello_world "4096 16384 32768 65536" "100"
for kf in 700 1000 2000 2800; do echo "== chunk bound $kf k env-frames"; CAMPX_SHAPE_CHUNK_KF=$kf bash tools/gpu_sweep.sh hello_world "32768 65536" "100"; done
