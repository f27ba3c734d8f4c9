R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "bucketed_grad_sync" 2>&1 | grep -E "passed|failed|rror|assert" | tail -5
for v in "X=1" "EBEN_COMM_STREAM=side"; do echo "== force-ddp $v"; for i in 1 2; do env $v python $R/bench.py --force-ddp --steps 30 --warmup 5 --no-cpu-baseline --no-f32-leg 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'], d['comm']['exposed_ms_per_step'], d['comm']['buckets'])"; done; done
echo "== plain"; python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-leg 2>&1 >/dev/null | tail -1
