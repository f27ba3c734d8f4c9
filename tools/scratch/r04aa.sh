R=$GRAFT_REPO_ROOT
for L in melgan.3 melgan.4; do
for v in "X=1" "EBEN_PR_MAX_ROWS=4096" "EBEN_PR_MAX_ROWS=4096 EBEN_TAP3_BM=64" "EBEN_PR_MAX_ROWS=4096 EBEN_TAP3_BM=96" "EBEN_PR_MAX_ROWS=4096 EBEN_TAP3_BIG_LDS_KB=110" "EBEN_PR_MAX_ROWS=4096 EBEN_TAP3_BIG_KS=100000" "EBEN_TAP3_BM=128" "EBEN_TAP3_BIG_LDS_KB=110"; do
  printf "%s %-55s " $L "$v"; env $v python $R/tools/layer_bench_bl.py --only $L --iters 10 2>&1 | grep "^$L" | awk -F'|' '{print $2 "|" $3}'
done; done
