R=$GRAFT_REPO_ROOT
for v in "EBEN_RUBL_SEG32=1024 EBEN_RUBL_BKT128=64" "EBEN_RUBL_SEG32=2048 EBEN_RUBL_BKT128=64" "EBEN_RUBL_SEG32=2048 EBEN_RUBL_SEG64=1024 EBEN_RUBL_BKT128=64" "EBEN_RUBL_SEG32=4096 EBEN_RUBL_BKT128=64"; do echo "== $v"; for i in 1 2; do env $v python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-f32-leg 2>&1 >/dev/null | tail -1; done; done
