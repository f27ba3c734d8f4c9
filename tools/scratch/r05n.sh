cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_gpu_bl.py -x -q -m gpu 2>&1 | tail -3 > $O/r05n_tests.txt; cat $O/r05n_tests.txt
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-f32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PR2 on ', {k:d[k] for k in ('value','ms_per_step','host_enqueue_ms_per_step') if k in d})"
EBEN_PR_BIG=0 timeout 600 python bench.py --no-cpu-baseline --no-f32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PR2 off', {k:d[k] for k in ('value','ms_per_step') if k in d})"
done
EBEN_BIG=0 timeout 600 python bench.py --no-cpu-baseline --no-f32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('BIG off', {k:d[k] for k in ('value','ms_per_step') if k in d})"
