R=$GRAFT_REPO_ROOT
for kb in 150 80 64 48 32; do echo "== EBEN_TAP3_SPLIT_LDS_KB=$kb"; EBEN_TAP3_SPLIT_LDS_KB=$kb python $R/tools/gen_fwd_timeline.py > /tmp/o.txt 2>&1; grep -E "tap3_kernel" /tmp/o.txt | awk '{printf "%s ", $3}'; grep span /tmp/o.txt; tail -2 /tmp/o.txt | head -1; done
