R=$GRAFT_REPO_ROOT
for v in "EBEN_OVERLAP_D_ADAM=0" "EBEN_OVERLAP_D_ADAM=1" "EBEN_OVERLAP_D_ADAM=0" "EBEN_OVERLAP_D_ADAM=1"; do echo "== $v"; env $v python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-leg 2>&1 >/dev/null | tail -1; done
