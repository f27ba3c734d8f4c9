cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 1500 python -m pytest tests/test_gpu_models.py -x -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error" | head -8
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-f32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('x3 on ', {k:d[k] for k in ('value','ms_per_step','host_enqueue_ms_per_step','graphs_replayed') if k in d})"
EBEN_RU_FWD_X3=0 timeout 600 python bench.py --no-cpu-baseline --no-f32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('x3 off', {k:d[k] for k in ('value','ms_per_step') if k in d})"
done
