R=$GRAFT_REPO_ROOT
for v in "X=1" "EBEN_PG_HIGH_PRIORITY=0"; do echo "== force-ddp $v"; for i in 1 2; do env $v python $R/bench.py --force-ddp --steps 30 --warmup 5 --no-cpu-baseline --no-f32-leg 2>/tmp/err.txt | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'], d['comm']['exposed_ms_per_step'], 'graphs', d['graphs_replayed'], 'host', d['host_enqueue_ms_per_step'])" || tail -5 /tmp/err.txt; done; done
python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-leg 2>&1 >/dev/null | tail -1
