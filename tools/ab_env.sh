# Scratch A/B harness: the bench step under environment / library variants, interleaved so that box drift cancels.
# Usage: bash tools/ab_env.sh   (prints ms/step per variant and repetition)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { # label, env assignments...
  local label=$1; shift
  local ms=$(env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-f32-leg 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])")
  echo "$label $ms"
}
for rep in 1 2; do
  run base X=1
  run devkernarg HIP_FORCE_DEV_KERNARG=1
  run minb4 EBEN_HIP_LIB=$R/vibravox_amd/lib/var_minb4/libeben_hip.so
done
