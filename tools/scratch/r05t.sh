cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 600 python bench.py --no-cpu-baseline --no-f32-leg --steps 5 --warmup 2 2>&1 | grep -v "NCCL WARN\|^$\|amdgpu.ids" | cut -c1-400 | tail -20
