cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "pqmf or generator_matches or config1" 2>&1 | grep -E "passed|failed|FAILED" | head -5
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "fir or pqmf" 2>&1 | grep -E "passed|failed|FAILED|deselected" | head -5
python tools/pqmf_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/r05x_pqmf.txt
EBEN_PQMF_SHUFFLE=0 python tools/pqmf_bench.py 2>&1 | grep -v amdgpu.ids | sed 's/^/LDS form: /'
for i in 1 2; do for RING in 3 2; do
echo "== RING $RING"; EBEN_BIG_RING=$RING timeout 300 python tools/layer_bench_bl.py --filter melgan --iters 20 2>&1 | grep "melgan.[345]" | awk -F'|' '{print $1 "|" $2 "|" $3}'
EBEN_BIG_RING=$RING timeout 600 python bench.py --no-cpu-baseline --no-f32-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step') if k in d}, d['roofline']['launch_ms'], d['roofline']['isolated']['launch_ms'])"
done; done
