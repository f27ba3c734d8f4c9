R=$GRAFT_REPO_ROOT
for v in "X=1" "EBEN_D_FWD_SPREAD=0" "EBEN_D_BWD_SPREAD=0" "EBEN_TAP3_LDS_KB=32" "EBEN_TAP3_LDS_KB=64" "EBEN_TAP3_LDS_KB=78" "X=2"; do echo "== $v"; env $v python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-leg 2>/tmp/err.txt >/dev/null; tail -1 /tmp/err.txt; done
