R=$GRAFT_REPO_ROOT
for grp in "2,2,3" "2,2,1,1,1" "1,1,1,1,1,1,1" "3,2,2" "2,1,1,1,1,1"; do echo "== EBEN_GEN_BWD_GROUPS=$grp"; for i in 1 2; do EBEN_GEN_BWD_GROUPS=$grp python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-leg 2>&1 >/dev/null | tail -1; done; done
