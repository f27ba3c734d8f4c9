cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
for round in 1 2; do for V in "" t4prio3 t4hb1 t4prio3hb1; do
  if [ -n "$V" ]; then export EBEN_HIP_LIB=$R/vibravox_amd/lib/var/libeben_$V.so; else unset EBEN_HIP_LIB; fi
  echo "== ${V:-base}"; timeout 300 python tools/layer_bench_bl.py --filter melgan --iters 20 2>&1 | grep "melgan.[345]" | awk -F'|' '{print $1 "|" $2 "|" $3}'
done; done
