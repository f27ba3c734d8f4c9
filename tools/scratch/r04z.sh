R=$GRAFT_REPO_ROOT
for L in pqmf0.1 pqmf0.2 pqmf0.3 pqmf0.4 pqmf0.5 pqmf0.6; do
for v in "X=1" "EBEN_BLDW_FN=4" "EBEN_BLDW_NSPLIT=32" "EBEN_BLDW_NSPLIT=64" "EBEN_BLDW_NSPLIT=128" "EBEN_BLDW_NSPLIT=512" "EBEN_BLDW_FN=4 EBEN_BLDW_NSPLIT=256"; do
  printf "%s %-40s " $L "$v"; env $v python $R/tools/layer_bench_bl.py --only $L --iters 20 2>&1 | grep "^$L" | awk -F'|' '{print $4}'
done; done
