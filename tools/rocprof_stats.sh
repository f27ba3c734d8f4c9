# rocprofv3 --kernel-trace --stats (CSV) of the bench command; the kernel summary is copied to gpurun_out/r01_rocprofv3_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s60 -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32-leg > $O/s60.json 2> $O/s60.err
find $O/s60 -name "*stats*" | head; ls $O/s60 | head
f=$(find $O/s60 -name "*kernel_stats.csv" | head -1); head -30 "$f"; cp "$f" $O/r01_rocprofv3_kernel_stats.csv
find $O/s60 -name "*kernel_trace.csv" -delete; du -sh $O/s60
