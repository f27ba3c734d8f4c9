R=$GRAFT_REPO_ROOT
for r in 2 3 4; do echo "== ring $r"; if [ $r = 2 ]; then L=""; else L="EBEN_HIP_LIB=$R/variants/ring$r/libeben_hip.so"; fi
env $L python $R/tools/gen_fwd_timeline.py > /tmp/o.txt 2>&1; grep -E "tap3_kernel" /tmp/o.txt | awk '{printf "%s ", $3}'; grep span /tmp/o.txt
env $L python $R/tools/gen_bwd_timeline.py > /tmp/o.txt 2>&1; grep span /tmp/o.txt
done
