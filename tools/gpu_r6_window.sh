cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6m
python tools/bench_variants.py 262144; python tools/bench_variants.py 65536
timeout 900 python -m pytest tests/test_random_pickups.py tests/test_api_sequences.py tests/test_example.py tests/test_bench_launcher.py -m gpu -q --maxfail=5 2>&1 | tail -4
for i in 1 2 3 4 5 6 7 8; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; r=d['roofline']
print('RUN ms %.4f kms %.4f ratio %.3f | loop %.0f open_seen %.0f launches %.0f log_wait %.0f sync %.0f | dev %.0f' % (d['ms_per_step'], r['kernel_ms'], d['ms_per_step']/r['kernel_ms'], c['window_loop_us'], c['window_open_seen_us'], c['window_launches_us'], c['window_log_wait_us'], c['window_synchronize_us'], c['window_launches_done_us']))"
done
