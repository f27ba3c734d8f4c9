R=$GRAFT_REPO_ROOT
run() { echo "== $1"; shift; env "$@" python $R/tools/gen_fwd_timeline.py 2>/dev/null | grep -E "gc_|span" | awk '{printf "%s ", $3} END {print ""}'; }
run base X=1
run noMFMA EBEN_HIP_LIB=$R/variants/g1/libeben_hip.so
run noXloads EBEN_HIP_LIB=$R/variants/g2/libeben_hip.so
run noW EBEN_HIP_LIB=$R/variants/g4/libeben_hip.so
run noStores EBEN_HIP_LIB=$R/variants/g8/libeben_hip.so
run noMFMA_noX EBEN_HIP_LIB=$R/variants/g3/libeben_hip.so
