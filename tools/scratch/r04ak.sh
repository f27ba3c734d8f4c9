R=$GRAFT_REPO_ROOT
run() { echo "== $1"; shift; env "$@" python $R/tools/gen_conv_bench.py 2>/dev/null | awk '{printf "%s ", $2} END {print ""}'; }
run default X=1
run rt2ct2 EBEN_GC_RT=2 EBEN_GC_CT=2
run rt2ct1 EBEN_GC_RT=2 EBEN_GC_CT=1
run rt1ct1 EBEN_GC_RT=1 EBEN_GC_CT=1
