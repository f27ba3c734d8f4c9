R=$GRAFT_REPO_ROOT
for pri in "0,0,-1" "0,0,0"; do
echo "== EBEN_AUX_PRIORITY=$pri"; EBEN_AUX_PRIORITY=$pri python $R/tools/phase_times.py 2>&1 | grep -E "generator forward|total|joined|backward"
EBEN_AUX_PRIORITY=$pri python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg 2>&1 >/dev/null | tail -1
done
