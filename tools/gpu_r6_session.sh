#!/bin/bash
# Round-6 GPU session (through gpurun): tests, the bench line, the shape tier at the driver's
# command (repeats + a kernel trace), the round's profile set.   tools/gpu_r6_session.sh <tag> [notest]
set -u
tag=$1
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$tag
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
if [ "${2:-}" != "notest" ]; then
  timeout 2400 python -m pytest tests -m gpu -q --maxfail=10 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
fi
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.log 2>$O/bench.err; echo "bench rc=$?"
python3 - <<PY
import json
d = json.loads(open('$O/bench.log').read().strip().splitlines()[-1])
print(json.dumps({k: d[k] for k in ('value', 'ms_per_step')}), json.dumps(d['config'], indent=0)[:6000])
print(json.dumps({k: v for k, v in d['roofline'].items() if 'note' not in k}))
PY
grep BENCH_DETAILS $O/bench.err | cut -c1-600
for i in 1 2 3; do
  timeout 300 python bench.py --game hello_world --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-group 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('HELLO ms_per_step %.4f kernel_ms %.4f median %.4f min %.4f frac %.4f ceiling %s' % (d['ms_per_step'], r['kernel_ms'], r['per_launch_ms']['median'], r['per_launch_ms']['min'], r['frac'], r['measured_write_ceiling_gbs']))"
done
mkdir -p $O/shape_trace
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/shape_trace -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --game hello_world --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-group > $GRAFT_REPO_ROOT/$O/shape_trace/trace.log 2>&1)
grep -h '^{' $O/shape_trace/trace.log | tail -1 > $O/shape_trace/bench_line.json
python3 tools/rocpd_summary.py $O/shape_trace > $O/shape_trace_summary.txt 2>&1
python3 tools/kernel_gaps.py $O/shape_trace/trace_results.db > $O/shape_gaps.txt 2>&1; tail -30 $O/shape_gaps.txt
find $O/shape_trace -name "*.db" -delete
head -14 $O/shape_trace_summary.txt | cut -c1-200
bash tools/gpu_profile_all.sh $tag boat_race wall_world sokoban hello_world
