R=$GRAFT_REPO_ROOT
for v in "X=1" "EBEN_RU_BL=0" "EBEN_FUSED_NORMS=0" "EBEN_DX_PR=0" "EBEN_PACK3C_MIN=0"; do echo "== force-ddp $v"; env $v python $R/bench.py --force-ddp --steps 30 --warmup 5 --no-cpu-baseline --no-f32-leg 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'], d['comm']['exposed_ms_per_step'], 'graphs', d['graphs_replayed'], 'host', d['host_enqueue_ms_per_step'], 'settle', d['settle_steps'])"; 
env $v python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-leg 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('plain', d['ms_per_step'], d['value'], 'graphs', d['graphs_replayed'], 'host', d['host_enqueue_ms_per_step'])"; done
