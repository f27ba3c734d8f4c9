R=$GRAFT_REPO_ROOT
run() { echo "== $1"; shift; env "$@" python $R/tools/gen_fwd_timeline.py 2>/dev/null | grep -E "tap3|space_to|span" | awk '{printf "%s %s | ", $3, $5} END {print ""}'; }
run A X=1
run B EBEN_HIP_LIB=$R/variants/B/libeben_hip.so
run C EBEN_HIP_LIB=$R/variants/C/libeben_hip.so
run A78 EBEN_TAP3_SPLIT_LDS_KB=78
run B78 EBEN_HIP_LIB=$R/variants/B/libeben_hip.so EBEN_TAP3_SPLIT_LDS_KB=78
