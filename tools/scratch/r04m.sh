R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O/r04m; cd /tmp; export TMPDIR=/tmp
for cap in 512 2048 8192; do
EBEN_PACK3_CAP=$cap rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04m_s -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-leg > $O/r04m/stats_bench_$cap.json 2> $O/r04m/stats.err; echo "cap $cap"; grep pack3 "$(find $O/r04m_s -name "*kernel_stats.csv" | head -1)"; rm -rf $O/r04m_s
EBEN_PACK3_CAP=$cap python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-leg 2>&1 >/dev/null | tail -1
done
