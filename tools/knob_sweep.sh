# A/B of scheduling / tile-budget knobs on one box: bash tools/knob_sweep.sh  (each line: knob setting, ms/step)
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { printf "%-44s " "$*"; env "$@" python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-leg 2>/dev/null | tail -1 | python -c "import sys,json; b=json.loads(sys.stdin.read()); print(b['ms_per_step'])"; }
run EBEN_NOP=1
for kv in "$@"; do run $kv; done
run EBEN_NOP=1
