cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 900 python tools/layer_bench_bl.py --iters 20 2>&1 | grep -v amdgpu.ids > $O/r05j_layers.txt; cat $O/r05j_layers.txt
timeout 600 python bench.py --no-cpu-baseline --no-f32-leg > $O/r05j_bench.json 2> $O/r05j_bench.err; cat $O/r05j_bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','host_enqueue_ms_per_step') if k in d}); print(d.get('roofline'))"
EBEN_BIG=0 timeout 600 python bench.py --no-cpu-baseline --no-f32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('EBEN_BIG=0', {k:d[k] for k in ('value','ms_per_step') if k in d})"
